/* Stand-in for Tracter's TracterObject.h (third party, absent): the one base class the reference's decoder classes derive from. */
#ifndef REFBASE_TRACTEROBJECT_H
#define REFBASE_TRACTEROBJECT_H
#include <stdlib.h>
#include <string>
namespace Tracter {
class Object {
public:
    Object() : mObjectName(0) {}
    virtual ~Object() throw() {}
protected:
    const char *mObjectName;
    int GetEnv(const char *suffix, int dflt) {
        std::string n = std::string(mObjectName ? mObjectName : "") + "_" + suffix;
        const char *e = getenv(n.c_str());
        return e ? atoi(e) : dflt;
    }
};
}
#endif
