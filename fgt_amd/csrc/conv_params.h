// Launch parameters shared by the conv kernels (implicit GEMM and direct).
#pragma once
#include "common.h"

struct ConvP {
    fgt_conv_desc d;
    const float *x0, *x1, *w, *cscale, *cbias, *aux1, *aux2;
    float* out;
    int M, HoWo, Cg0, Cg1, Cg, K, Cout_g, Hin, Win, nk;
};

// conv_direct.hip
bool fgt_conv_direct_eligible(const ConvP& p);
int fgt_conv_direct(const ConvP& p, hipStream_t s);
