// placeholder until the MLP kernels land
#include <string>
#include "lde_device.h"
namespace lde {
struct MlpPlan { int dummy; };
int mlp_plan_create(const lde_problem_desc&, MlpPlan**, std::string& err) { err = "MLP RHS not built yet"; return LDE_ERR_UNSUPPORTED; }
void mlp_plan_destroy(MlpPlan*) {}
int mlp_reserve(MlpPlan*, int, int, std::string&) { return LDE_ERR_UNSUPPORTED; }
int mlp_set_weights(MlpPlan*, const float*, hipStream_t, std::string&) { return LDE_ERR_UNSUPPORTED; }
int mlp_forward(MlpPlan*, const float*, const float*, const float*, const double*, const KOpts&, float*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, hipStream_t, std::string&) { return LDE_ERR_UNSUPPORTED; }
int mlp_adjoint(MlpPlan*, const float*, const float*, const float*, const double*, const KOpts&, const float*, float*, float*, float*, int32_t*, int32_t*, int32_t*, int32_t*, hipStream_t, std::string&) { return LDE_ERR_UNSUPPORTED; }
}
