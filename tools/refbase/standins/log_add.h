/* Stand-in for Torch3's log_add.h (see general.h in this directory): the constants the reference uses, and logAdd (unused on the flat path). */
#ifndef REFBASE_LOG_ADD_H
#define REFBASE_LOG_ADD_H
#include "general.h"
#define LOG_2_PI 1.83787706640934548355
#define LOG_ZERO (-INF)
#define LOG_ONE 0
namespace Torch {
inline real logAdd(real x, real y)
{
    if (x < y) { real t = x; x = y; y = t; }
    real d = y - x;
    if (d < -18.42) return x;
    return x + (real)log1p(exp((double)d));
}
}
#endif
