// Walk-kernel instantiations: the kernels that read every flag at run time, and the reference's production
// configuration (MC code.py: reflecting floor, Henyey-Greenstein, two escape tests, no depth bins) on all three random
// streams, with and without rand_mu / monochromatic fixed off.
#include "rt_walk.hpp"

namespace rt {

#ifdef RT_ABLATION
extern "C" int rt_debug_wave_trace(unsigned long long* out, int n_waves) {  // tools/wave_trace.py
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_trace), sizeof(unsigned long long) * 8 * (size_t)n_waves);
}
#endif

void register_walk_base(std::vector<WalkVariantInfo>& out) {
    add_walk_variant<Runtime, RT_RNG_PHILOX52, false>(out);    // kWalkRuntime
    add_walk_variant<Runtime, RT_RNG_PHILOX32X4, false>(out);  // kWalkRuntime + RT_RNG_PHILOX32X4
    add_walk_variant<Runtime, RT_RNG_PHILOX40, false>(out);    // kWalkRuntime + RT_RNG_PHILOX40
    add_walk_variant<FixedBase, RT_RNG_PHILOX52, true>(out);   // the headline kernel
    add_walk_variant<FixedBase, RT_RNG_PHILOX52, false>(out);
    add_walk_variant<FixedBase, RT_RNG_PHILOX32X4, true>(out);
    add_walk_variant<FixedBase, RT_RNG_PHILOX32X4, false>(out);
    add_walk_variant<FixedBase, RT_RNG_PHILOX40, true>(out);
    add_walk_variant<FixedBase, RT_RNG_PHILOX40, false>(out);
}

}  // namespace rt
