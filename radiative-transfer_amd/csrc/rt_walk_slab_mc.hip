// Walk-kernel instantiations: the slab in the loop order of MC code.py, every lower boundary (CheckBoundary,
// MC code.py:522-574) x {Henyey-Greenstein, scatter='iso'} x {no depth bins, depth bins}.  (Reflecting floor +
// Henyey-Greenstein without depth bins is the production configuration, registered with the base group.)
#include "rt_walk.hpp"

namespace rt {

template <int B, int L>
static void add_pair(std::vector<WalkVariantInfo>& out) {
    if constexpr (!(B == RT_BOUNDARY_REFLECT && L == RT_SCATTER_HG)) add_walk_variant<Fixed<RT_ORDER_MC, B, L, false>>(out);
    add_walk_variant<Fixed<RT_ORDER_MC, B, L, true>>(out);
    // Henyey-Greenstein, the law of every production run: also with rand_mu and monochromatic fixed off (the
    // reference's defaults), like the production configuration itself -- the depth-resolved deposition of the default
    // run is one of these
    if constexpr (L == RT_SCATTER_HG) {
        if constexpr (B != RT_BOUNDARY_REFLECT) add_walk_variant<Fixed<RT_ORDER_MC, B, L, false>, RT_RNG_PHILOX52, true>(out);
        add_walk_variant<Fixed<RT_ORDER_MC, B, L, true>, RT_RNG_PHILOX52, true>(out);
    }
}

void register_walk_slab_mc(std::vector<WalkVariantInfo>& out) {
    add_pair<RT_BOUNDARY_REFLECT, RT_SCATTER_HG>(out);
    add_pair<RT_BOUNDARY_REFLECT, RT_SCATTER_ISO>(out);
    add_pair<RT_BOUNDARY_ABSORB, RT_SCATTER_HG>(out);
    add_pair<RT_BOUNDARY_ABSORB, RT_SCATTER_ISO>(out);
    add_pair<RT_BOUNDARY_REEMIT, RT_SCATTER_HG>(out);
    add_pair<RT_BOUNDARY_REEMIT, RT_SCATTER_ISO>(out);
}

}  // namespace rt
