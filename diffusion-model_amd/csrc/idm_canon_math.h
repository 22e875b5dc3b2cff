// Canonical float32 elementary functions shared by the parity-mode translation units
// (which are compiled with -ffp-contract=off): every step is an explicitly rounded operation,
// so a scalar C implementation of the same formulas gives the same bits.
#pragma once
#include "idm_device.h"

#include <math.h>

DEVI float cexp(float x)
{
    x = fminf(fmaxf(x, -87.0f), 87.0f);
    const float n = rintf(__fmul_rn(x, 1.44269504088896341f));
    float r = __fmaf_rn(n, -0.693359375f, x);
    r = __fmaf_rn(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __fmaf_rn(p, r, 1.3981999507e-3f);
    p = __fmaf_rn(p, r, 8.3334519073e-3f);
    p = __fmaf_rn(p, r, 4.1665795894e-2f);
    p = __fmaf_rn(p, r, 1.6666665459e-1f);
    p = __fmaf_rn(p, r, 5.0000001201e-1f);
    const float r2 = __fmul_rn(r, r);
    p = __fmaf_rn(p, r2, r);
    p = __fadd_rn(p, 1.0f);
    return ldexpf(p, (int)n);
}

DEVI float csilu(float y) { return __fdiv_rn(y, __fadd_rn(1.0f, cexp(-y))); }

