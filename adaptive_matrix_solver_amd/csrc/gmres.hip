// Batched restarted GMRES (placeholder translation unit; filled in by the GMRES milestone).
#include "common.h"
#include "../../include/maus_hip.h"
hipStream_t maus_ctx_stream(maus_ctx* c);
void maus_ctx_set_error(maus_ctx* c, const char* m);

int maus_gmres_run(maus_ctx* ctx, const int*, int, const double*, const double*, int, const int32_t*, double, int, int,
                   int32_t*, int32_t*, int32_t*) {
    maus_ctx_set_error(ctx, "maus_gmres: not built yet");
    return -2;
}
int maus_jacobi_check_run(maus_ctx* ctx, int, const double*, const double*, int32_t*) {
    maus_ctx_set_error(ctx, "maus_jacobi_check: not built yet");
    return -2;
}
