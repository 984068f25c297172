// Walk-kernel instantiations: the slab in the gray script's loop order (Numpy_random_walk.py:313-396: boundary before
// the single escape test, momentum weighted by lambda), every boundary x {hg, iso} x {no depth bins, depth bins}.
#include "rt_walk.hpp"

namespace rt {

template <int B, int L>
static void add_pair(std::vector<WalkVariantInfo>& out) {
    add_walk_variant<Fixed<RT_ORDER_GRAY, B, L, false>>(out);
    add_walk_variant<Fixed<RT_ORDER_GRAY, B, L, true>>(out);
}

void register_walk_slab_gray(std::vector<WalkVariantInfo>& out) {
    add_pair<RT_BOUNDARY_REFLECT, RT_SCATTER_HG>(out);
    add_pair<RT_BOUNDARY_REFLECT, RT_SCATTER_ISO>(out);
    add_pair<RT_BOUNDARY_ABSORB, RT_SCATTER_HG>(out);
    add_pair<RT_BOUNDARY_ABSORB, RT_SCATTER_ISO>(out);
    add_pair<RT_BOUNDARY_REEMIT, RT_SCATTER_HG>(out);
    add_pair<RT_BOUNDARY_REEMIT, RT_SCATTER_ISO>(out);
}

}  // namespace rt
