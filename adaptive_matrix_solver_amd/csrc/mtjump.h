// Tap lists of the polynomials one MT19937 jump launch may apply (mtdev.hip, capi.hip): entry v = x^(v J), v = 1 alone for a
// uniform launch, v = 1..15 for a launch that lifts one hexadecimal digit of the draw index (mtplan.cpp).
#pragma once
struct MausJumpPolys { const int* taps[16]; int ntap16[16]; };
