/* Stand-in for Torch3's general.h (third party, absent from the reference tree and from this image) - written from what the
 * reference's sources use of it; NOT reference code and NOT Torch3 code.  Only tools/refbase uses it: the
 * build it enables is a TIMING and differential aid, it pins nothing (a build against stand-ins is not a reference build). */
#ifndef REFBASE_GENERAL_H
#define REFBASE_GENERAL_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <float.h>
#include <math.h>
#include <algorithm>
#define real float
#define INF FLT_MAX
namespace Torch {
inline void error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); fprintf(stderr, "error: "); vfprintf(stderr, fmt, ap); fprintf(stderr, "\n"); va_end(ap); exit(1); }
inline void warning(const char *fmt, ...) { va_list ap; va_start(ap, fmt); fprintf(stderr, "warning: "); vfprintf(stderr, fmt, ap); fprintf(stderr, "\n"); va_end(ap); }
inline void message(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); fprintf(stderr, "\n"); va_end(ap); }
}
using std::min;
using std::max;
#endif
