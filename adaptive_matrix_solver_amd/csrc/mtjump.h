// Tap lists of the polynomials one MT19937 jump launch may apply (mtdev.hip, capi.hip): entry v = x^(v J), v = 1 alone for a
// uniform launch, v = 1..15 for a launch that lifts one hexadecimal digit of the draw index (mtplan.cpp).
#pragma once
// taps[v]: nlo16[v] groups of 16 taps below the window split (maus_mt_tap_split), then the others minus the split, ntap16[v] groups in all
struct MausJumpPolys { const int* taps[16]; int ntap16[16]; int nlo16[16]; };
