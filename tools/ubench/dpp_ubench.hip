// Micro-benchmark / semantics check for fp64 DPP row_newbcast on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
template <int K> __device__ __forceinline__ double rb(double v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xf, 0xf, true);
}
// broadcast of lane K (< 16) of each 32-lane half-wavefront to its 32 lanes: every 16-lane row takes its own lane K
// (row_newbcast), then v_permlane16_swap copies the even row's value into the odd row of the pair (per dword)
template <int K> __device__ __forceinline__ double bc32(double v) {
    // (32-bit DPP moves: splitting the result of the 64-bit one makes this LLVM emit V_MOV_B64_dpp with an undef tied
    // half and its machine verifier rejects it at COMPILE time: "Illegal instruction detected: Operand has incorrect
    // register class")
    const unsigned lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + K, 0xf, 0xf, true);
    const unsigned hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + K, 0xf, 0xf, true);
    auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)h[0], (int)l[0]);
}
__device__ __forceinline__ double bp32(double v, int k) {
    const int src = ((threadIdx.x & 32) + k) * 4;
    const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__global__ void sem(double* out, const double* in) {
    int l = threadIdx.x;
    double x = in[l];
    out[l] = rb<3>(x);
    out[64 + l] = rb<15>(x);
    double a = 1.0, m = 2.0;
    // asm fused: a += bcast5(x) * m
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(x), "v"(m));
    out[128 + l] = a;
    out[192 + l] = bc32<3>(x);
    out[256 + l] = bp32(x, 3);
}
#define REP16(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
template <int MODE>
__global__ __launch_bounds__(64) void bench(double* out, const double* in, int iters) {
    int l = threadIdx.x;
    double x = in[l], m = in[l + 64];
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // plain fma, 4 independent accumulators
            REP16(a0 = fma(m, x, a0); a1 = fma(m, x, a1); a2 = fma(m, x, a2); a3 = fma(m, x, a3);)
        } else if (MODE == 1) {  // builtin dpp mov + fma
            REP16(a0 = fma(m, rb<1>(x), a0); a1 = fma(m, rb<2>(x), a1); a2 = fma(m, rb<3>(x), a2); a3 = fma(m, rb<4>(x), a3);)
        } else if (MODE == 2) {  // fused asm fmac dpp (x written long ago: no hazard)
            REP16(asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fmac_f64_dpp %1, %4, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fmac_f64_dpp %2, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fmac_f64_dpp %3, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(m));)
        } else if (MODE == 3) {  // dependent chain fma
            REP16(a0 = fma(m, a0, x); a0 = fma(m, a0, x); a0 = fma(m, a0, x); a0 = fma(m, a0, x);)
        } else if (MODE == 4) {  // dependent chain through dpp mov + fma (solve-like)
            REP16(a0 = fma(m, rb<1>(a0), x); a0 = fma(m, rb<2>(a0), x); a0 = fma(m, rb<3>(a0), x); a0 = fma(m, rb<4>(a0), x);)
        } else if (MODE == 5) {  // readlane-based broadcast chain (v1 style)
            REP16({int lo=__builtin_amdgcn_readlane(__double2loint(a0),3), hi=__builtin_amdgcn_readlane(__double2hiint(a0),3); a0 = fma(m, __hiloint2double(hi,lo), x);}
                  {int lo=__builtin_amdgcn_readlane(__double2loint(a0),5), hi=__builtin_amdgcn_readlane(__double2hiint(a0),5); a0 = fma(m, __hiloint2double(hi,lo), x);}
                  {int lo=__builtin_amdgcn_readlane(__double2loint(a0),7), hi=__builtin_amdgcn_readlane(__double2hiint(a0),7); a0 = fma(m, __hiloint2double(hi,lo), x);}
                  {int lo=__builtin_amdgcn_readlane(__double2loint(a0),9), hi=__builtin_amdgcn_readlane(__double2hiint(a0),9); a0 = fma(m, __hiloint2double(hi,lo), x);})
        } else if (MODE == 6) {  // rcp chain
            REP16(a0 = 1.0 / (a0 + x); a0 = 1.0 / (a0 + x); a0 = 1.0 / (a0 + x); a0 = 1.0 / (a0 + x);)
        } else if (MODE == 7) {  // 16-lane layout, two register slots: ONE row_newbcast feeds TWO fmas (what the kernels do)
            REP16({double b = rb<1>(x); a0 = fma(m, b, a0); a1 = fma(m, b, a1);} {double b = rb<2>(x); a2 = fma(m, b, a2); a3 = fma(m, b, a3);}
                  {double b = rb<3>(x); a0 = fma(m, b, a0); a1 = fma(m, b, a1);} {double b = rb<4>(x); a2 = fma(m, b, a2); a3 = fma(m, b, a3);})
        } else if (MODE == 8) {  // 32-lane group broadcast (lane j < 16 of each half-wavefront): row_newbcast + v_permlane16_swap
                                 // per dword, feeding ONE fma (a 32-lane layout has one register slot)
            REP16(a0 = fma(m, bc32<1>(x), a0); a1 = fma(m, bc32<2>(x), a1); a2 = fma(m, bc32<3>(x), a2); a3 = fma(m, bc32<4>(x), a3);)
        } else if (MODE == 9) {  // the same through the LDS crossbar: ds_bpermute_b32 x 2 + fma
            REP16(a0 = fma(m, bp32(x, 1), a0); a1 = fma(m, bp32(x, 2), a1); a2 = fma(m, bp32(x, 3), a2); a3 = fma(m, bp32(x, 4), a3);)
        }
    }
    out[blockIdx.x * 64 + l] = a0 + a1 + a2 + a3;
}
template <int MODE> void run(const char* name, int waves_per_simd) {
    int blocks = 1024 * waves_per_simd, iters = 200;
    double *out, *in; hipMalloc(&out, blocks * 64 * 8); hipMalloc(&in, 128 * 8);
    std::vector<double> h(128, 1e-3); hipMemcpy(in, h.data(), 128 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    bench<MODE><<<blocks, 64>>>(out, in, iters); hipDeviceSynchronize();
    hipEventRecord(e0); bench<MODE><<<blocks, 64>>>(out, in, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = 64.0 * iters;  // fma-ish ops per wave
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.1f ns per op per wave (%.1f cycles @2.4GHz)\n", name, waves_per_simd, ms, ms * 1e6 / ops, ms * 1e6 / ops * 2.4);
    hipFree(out); hipFree(in);
}
int main() {
    double *out, *in; hipMalloc(&out, 320 * 8); hipMalloc(&in, 64 * 8);
    std::vector<double> h(64); for (int i = 0; i < 64; ++i) h[i] = 100 + i;
    hipMemcpy(in, h.data(), 64 * 8, hipMemcpyHostToDevice);
    sem<<<1, 64>>>(out, in); std::vector<double> o(320); hipMemcpy(o.data(), out, 320 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        double e3 = 100 + (l / 16) * 16 + 3, e15 = 100 + (l / 16) * 16 + 15, e5 = 1.0 + (100 + (l / 16) * 16 + 5) * 2.0;
        if (o[l] != e3 || o[64 + l] != e15 || o[128 + l] != e5) { bad++; if (bad < 5) printf("lane %d: %g %g %g expected %g %g %g\n", l, o[l], o[64+l], o[128+l], e3, e15, e5); }
    }
    printf("semantics: %s\n", bad ? "MISMATCH" : "row_newbcast OK (mov_b64_dpp and fmac_f64_dpp)");
    int bad32 = 0;
    for (int l = 0; l < 64; ++l) {
        const double e = 100 + (l / 32) * 32 + 3;
        if (o[192 + l] != e || o[256 + l] != e) { bad32++; if (bad32 < 5) printf("lane %d: bc32 %g bpermute %g expected %g\n", l, o[192 + l], o[256 + l], e); }
    }
    printf("32-lane group broadcast: %s\n", bad32 ? "MISMATCH" : "OK (row_newbcast + v_permlane16_swap; ds_bpermute)");
    for (int w = 1; w <= 2; ++w) {
        run<0>("fma x4 independent", w); run<1>("mov_dpp+fma x4 indep", w); run<2>("fmac_dpp(asm) x4 indep", w);
        run<7>("mov_dpp + 2 fma (16-lane, 2 slots)", w); run<8>("bcast32 dpp+permlane16_swap + fma", w); run<9>("bcast32 ds_bpermute x2 + fma", w);
        run<3>("fma dependent chain", w); run<4>("mov_dpp+fma dep chain", w); run<5>("readlane x2+fma dep chain", w); run<6>("add+div dep chain", w);
    }
    return 0;
}
