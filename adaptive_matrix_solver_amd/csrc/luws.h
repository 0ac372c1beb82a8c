// LU workspace descriptor shared by the C-ABI layer (capi.hip) and the LU driver (lu.hip).
#pragma once
#include "common.h"

struct LuWs {
    c128* H; long ldh; long strideH; int n; int npad; int G;
    // implicit pivoting: rows never move.  perm[g][i] = physical row of H that holds logical row i; the finished rows of
    // U (and the carried right-hand side) are written in LOGICAL order to the second array U (same ld / stride as H)
    c128* U; int* perm;
    void* mw_sync = nullptr;     // per-matrix rendezvous area of the multi-workgroup panel; null unless this LU is alone on the device
    int* ipiv; int* info; int* flags;
    hipStream_t st;
    void (*tick)(void* ud, int klass, int phase, double flops, double bytes); void* ud;
};
