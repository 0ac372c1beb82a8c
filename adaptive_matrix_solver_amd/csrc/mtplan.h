// Host-side plan of the device MT19937 regeneration (MAUS_PERT_MT19937): which generator start states a sub-batch
// needs and how they are reached from the one NumPy state by binary lifting.  Pure host arithmetic -- no HIP -- so that
// it builds under the CPU sanitizers together with mt19937.cpp (`make asan`).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include "../../include/maus_hip.h"

struct MausMtPlan {
    int S = 1;                 // sub-streams per rand(N,N) draw (one workgroup each)
    uint64_t E = 0;            // elements per sub-stream
    int ngen = 0;              // generators = 2 draws x g candidates x S sub-streams
    uint64_t dj = 0, dj2 = 0;  // jump strides in blocks of 624 words: per draw, per sub-stream
    // staging image for the device: extra[ngen] (real block regenerations after the jumps) | rpos[ngen] (position in the
    // block) | the selection list of every lifting level
    std::vector<int> hs;
    // states[hs[off + i]] <- x^(mult_i * J) applied to states[hs[off + i] - src_off]   (src_off = 0: in place).
    // multi: hs[off + count + i] = mult_i in 1..15 (one hexadecimal digit of the draw index per level); else mult_i = 1
    struct Level { size_t off; int count; uint64_t J; int src_off; bool multi; };
    std::vector<Level> levels;
};

// Candidates [first, first+g) of a run described by d (n x n matrices).  s_override > 0 forces the sub-stream count.
// Returns 0, or -1 with *err set.
int maus_mt_plan(const maus_mt_desc* d, int n, int first, int g, int s_override, MausMtPlan* out, const char** err);
