// placeholder until the backward kernels land
#include "common.hpp"
#include "mlp_layout.hpp"
using namespace nerfhip;
extern "C" {
int64_t nerf_mlp_scratch_floats(int use_viewdirs, int64_t n_pts) { return nerflayout::act_floats(use_viewdirs, n_pts); }
int nerf_mlp_bwd(int64_t, int, const float*, const float*, int, int, const float*, const float*, float*, float*, void*) {
    return fail(NERF_E_ARG, "nerf_mlp_bwd: not implemented yet");
}
}
