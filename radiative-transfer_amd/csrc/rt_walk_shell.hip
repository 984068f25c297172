// Walk-kernel instantiations: the spherical shell (RT_GEOM_SHELL, no counterpart in the reference) in the MC loop
// order, every treatment of the cavity x {Henyey-Greenstein, exact Draine law} x {no radial bins, radial bins}.
#include "rt_walk.hpp"

namespace rt {

template <int B, int L>
static void add_pair(std::vector<WalkVariantInfo>& out) {
    add_walk_variant<Fixed<RT_ORDER_MC, B, L, false, false, true>>(out);
    add_walk_variant<Fixed<RT_ORDER_MC, B, L, true, false, true>>(out);
}

void register_walk_shell(std::vector<WalkVariantInfo>& out) {
    add_pair<RT_BOUNDARY_REFLECT, RT_SCATTER_HG>(out);
    add_pair<RT_BOUNDARY_REFLECT, RT_SCATTER_DRAINE_EXACT>(out);
    add_pair<RT_BOUNDARY_ABSORB, RT_SCATTER_HG>(out);
    add_pair<RT_BOUNDARY_ABSORB, RT_SCATTER_DRAINE_EXACT>(out);
    add_pair<RT_BOUNDARY_REEMIT, RT_SCATTER_HG>(out);
    add_pair<RT_BOUNDARY_REEMIT, RT_SCATTER_DRAINE_EXACT>(out);
}

}  // namespace rt
