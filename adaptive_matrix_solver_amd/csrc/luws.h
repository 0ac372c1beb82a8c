// LU workspace descriptor shared by the C-ABI layer (capi.hip) and the LU driver (lu.hip).
#pragma once
#include "common.h"

struct LuWs {
    c128* H; long ldh; long strideH; int n; int npad; int G;
    int* ipiv; int* info; int* flags;
    hipStream_t st;
    void (*tick)(void* ud, int klass, int phase, double flops, double bytes); void* ud;
};
