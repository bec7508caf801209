// kernels_direct.hip: band Cholesky on the device (femcy_direct_solve)
#pragma once
#include "ctx.hpp"

namespace femcy {

int direct_solve(Ctx* c, const double* d_b, double* d_x, femcy_direct_info* info);
int direct_plan(Ctx* c, femcy_direct_info* info);   // femcy_direct_plan: the band femcy_direct_solve would factor
int direct_set_max_bytes(Ctx* c, int64_t bytes);
int direct_set_update_variant(Ctx* c, int64_t v);   // FEMCY_TUNE_DIRECT_UPDATE
void direct_release(Ctx* c);   // femcy_ctx_destroy

}  // namespace femcy
