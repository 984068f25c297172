// Walk-kernel instantiations: the remaining options of the slab, each with the production flags otherwise -- the exact
// Draine law (every boundary, with and without depth bins), abs_only (MC code.py:478-481) and the per-photon
// escape-cosine list (RunPhotons' escaped_mu, MC code.py:693,779; both loop orders).
#include "rt_walk.hpp"

namespace rt {

void register_walk_options(std::vector<WalkVariantInfo>& out) {
    add_walk_variant<Fixed<RT_ORDER_MC, RT_BOUNDARY_REFLECT, RT_SCATTER_DRAINE_EXACT, false>>(out);
    add_walk_variant<Fixed<RT_ORDER_MC, RT_BOUNDARY_REFLECT, RT_SCATTER_DRAINE_EXACT, true>>(out);
    add_walk_variant<Fixed<RT_ORDER_MC, RT_BOUNDARY_ABSORB, RT_SCATTER_DRAINE_EXACT, false>>(out);
    add_walk_variant<Fixed<RT_ORDER_MC, RT_BOUNDARY_ABSORB, RT_SCATTER_DRAINE_EXACT, true>>(out);
    add_walk_variant<Fixed<RT_ORDER_MC, RT_BOUNDARY_REEMIT, RT_SCATTER_DRAINE_EXACT, false>>(out);
    add_walk_variant<Fixed<RT_ORDER_MC, RT_BOUNDARY_REEMIT, RT_SCATTER_DRAINE_EXACT, true>>(out);
    add_walk_variant<Fixed<RT_ORDER_MC, RT_BOUNDARY_REFLECT, RT_SCATTER_HG, false, true>>(out);   // abs_only
    add_walk_variant<Fixed<RT_ORDER_MC, RT_BOUNDARY_REFLECT, RT_SCATTER_HG, true, true>>(out);
    add_walk_variant<Fixed<RT_ORDER_MC, RT_BOUNDARY_REFLECT, RT_SCATTER_HG, false, false, false, true>>(out);    // escaped_mu
    add_walk_variant<Fixed<RT_ORDER_GRAY, RT_BOUNDARY_REFLECT, RT_SCATTER_HG, false, false, false, true>>(out);
}

}  // namespace rt
