// LU workspace descriptor shared by the C-ABI layer (capi.hip) and the LU driver (lu.hip).
#pragma once
#include "common.h"

struct LuWs {
    c128* H; long ldh; long strideH; int n; int npad; int G;
    // implicit pivoting: rows never move.  perm[g][i] = physical row of H that holds logical row i; the finished rows of
    // U (and the carried right-hand side) are written in LOGICAL order to the second array U (same ld / stride as H)
    c128* U; int* perm;
    void* mw_sync = nullptr;     // per-matrix rendezvous area of the multi-workgroup panel; null unless this LU is alone on the device
    // multi-workgroup panel: how long a rendezvous may wait (100 MHz ticks) before the matrix is reported as
    // info = INT_MIN (the caller then repeats the batch with one workgroup per matrix), and the test hook that makes one
    // workgroup skip an arrival (MAUS_PANEL_MW_FORCE_ABORT)
    unsigned long long mw_timeout = 200000000ull;
    int mw_force_abort = 0;
    int* ipiv; int* info; int* flags;
    hipStream_t st;
    void (*tick)(void* ud, int klass, int phase, double flops, double bytes); void* ud;
};
