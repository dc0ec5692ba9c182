// Compile-only: the adapter's in-tree branch (Juicer's Decoder.h included first).
#include "Decoder.h"
#include "juicer_amd_decoder.hpp"
#include <type_traits>
static_assert(std::is_base_of<Juicer::IDecoder, JuicerAmd::GpuWFSTDecoder>::value, "must derive from Juicer::IDecoder");
static_assert(std::is_same<JuicerAmd::DecHyp, Juicer::DecHyp>::value, "must return Juicer::DecHyp");
Juicer::IDecoder *make(const jd_net *n, const jd_am *a)
{
    return new JuicerAmd::GpuWFSTDecoder(n, a, 0.0f, 150.0f, 0.0f, 0.0f, 0);
}
static_assert(std::is_base_of<Juicer::IDecoder, JuicerAmd::GpuWFSTOnTheFlyDecoder>::value, "the on-the-fly mirror is an IDecoder too");
Juicer::IDecoder *make_on_the_fly(const jd_net *cl, const jd_net *g, const jd_am *a)
{
    return new JuicerAmd::GpuWFSTOnTheFlyDecoder(cl, g, a, 150.0f, 0.0f, 0, true);
}
