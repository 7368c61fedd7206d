// valu_calibrate.hip -- what one SIMD of this GPU really issues per clock, measured (VERDICT r3 #3a).
//
// bench.py's `roofline_issue` divides the VALU lane-cycles the refinement kernel used by what the machine offers;
// "what the machine offers" was an assumption (16 lanes per SIMD per clock, a wave64 instruction = 4 cycles) that
// /opt/skills/guides/MI355X_MICROARCH.md contradicts (SIMD-32, 2-cycle issue).  This program times dense loops of the
// instruction kinds the kernel is made of, with 1 and 2 wavefronts per SIMD (the kernel runs at 2):
//     v_fma_f32           independent        (the plain float32 rate)
//     v_pk_fma_f32        independent        (packed pairs: texel_pair, the red/green mean chains)
//     v_add_f32 ... dpp   independent        (the hand-over adds of the lane chains)
//     v_fma_f64           independent        (the optimiser's float64 algebra)
//     v_add_f32           ONE dependent chain (the sequential sums: latency, not rate)
//     v_add_f32 dpp       ONE dependent chain
//     v_fma_f64           ONE dependent chain
//     v_cvt_f32_ubyte0    independent        (texel conversion)
// and a streaming read through global_load_lds (16 B per lane, the page-in pattern) whose byte count is known, for the
// FETCH_SIZE calibration (run under rocprofv3 --pmc FETCH_SIZE: tools/pmc_calibrate.py holds the torch variants).
//
//     hipcc -O3 --offload-arch=gfx950 tools/valu_calibrate.hip -o tools/valu_calibrate && tools/valu_calibrate > profiles/calibration/valu_rates.json
//
// Output: one JSON object; per test the cycles (s_memtime, the shader clock) and nanoseconds (HIP events) per
// instruction of a wavefront, and lanes per clock per SIMD = 64 * waves_per_simd / cycles_per_instruction.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

constexpr int kChains = 16;   // independent accumulators per lane (a dependent instruction is never the next one issued)
constexpr int kIters = 4096;  // x 16 instructions per iteration

enum Kind { FMA32 = 0, PKFMA32, ADD32_DPP, FMA64, ADD32_CHAIN, ADD32_DPP_CHAIN, FMA64_CHAIN, CVT_UBYTE, N_KINDS };
static const char* kNames[N_KINDS] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32_dpp", "v_fma_f64", "v_add_f32 dependent chain",
                                      "v_add_f32_dpp dependent chain", "v_fma_f64 dependent chain", "v_cvt_f32_ubyte0"};

template <int KIND>
__global__ void __launch_bounds__(512) rate_kernel(float* out, unsigned long long* cycles, float seed) {
    float a[kChains];
    double d[kChains];
    v2f p[kChains];
    unsigned u[kChains];
    for (int i = 0; i < kChains; i++) { a[i] = seed + i; d[i] = seed + i; p[i] = v2f{seed + i, seed - i}; u[i] = (unsigned)(threadIdx.x * 7 + i); }
    const float b = seed * 0.5f, c = seed * 0.25f;
    const double bd = seed * 0.5, cd = seed * 0.25;
    const v2f bp = {b, c}, cp = {c, b};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int i = 0; i < kChains; i++) {
            if (KIND == FMA32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == PKFMA32) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(bp), "v"(cp));
            if (KIND == ADD32_DPP) a[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x138, 0xf, 0xf, true)) + b;
            if (KIND == FMA64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(bd), "v"(cd));
            if (KIND == ADD32_CHAIN) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[0]) : "v"(b));
            if (KIND == ADD32_DPP_CHAIN) a[0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[0]), 0x138, 0xf, 0xf, true)) + b;
            if (KIND == FMA64_CHAIN) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[0]) : "v"(bd), "v"(cd));
            if (KIND == CVT_UBYTE) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[i]) : "v"(u[i]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int i = 0; i < kChains; i++) s += a[i] + (float)d[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// one workgroup streams `bytes_per_wg` from global memory into LDS with global_load_lds_dwordx4 (16 B per lane, 1 KiB per
// wavefront instruction): the access pattern of the refinement kernel's page-in
__global__ void __launch_bounds__(256) lds_dma_stream(const uint4* src, size_t chunks_per_wave, unsigned* sink) {
    __shared__ uint4 buf[4 * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint4* g = src + ((size_t)blockIdx.x * 4 + wave) * chunks_per_wave * 64 + lane;
    auto lds = (__attribute__((address_space(3))) void*)(buf + wave * 64);
    for (size_t c = 0; c < chunks_per_wave; c++) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + c * 64), lds, 16, 0, 0);
        if ((c & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = buf[0].x;
}

template <int KIND>
static void run(int n_cus, int waves_per_simd, float* dout, unsigned long long* dcyc, std::string& json, bool first) {
    const int threads = 256 * waves_per_simd;  // 4 SIMDs per CU: 4 or 8 wavefronts per workgroup, one workgroup per CU
    const int blocks = n_cus;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, dout, dcyc, 1.0f);  // warm-up
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, dout, dcyc, 1.0f);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const int n_waves = blocks * threads / 64;
    std::vector<unsigned long long> cyc(n_waves);
    CHK(hipMemcpy(cyc.data(), dcyc, sizeof(unsigned long long) * n_waves, hipMemcpyDeviceToHost));
    double mean = 0;
    unsigned long long mx = 0;
    for (auto v : cyc) { mean += (double)v; mx = v > mx ? v : mx; }
    mean /= n_waves;
    const double instrs = (double)kIters * kChains;
    const double cpi = mean / instrs;  // cycles per instruction of ONE wavefront
    char buf[1024];
    snprintf(buf, sizeof buf,
             "%s\n    {\"instruction\": \"%s\", \"waves_per_simd\": %d, \"instructions_per_wave\": %.0f, \"cycles_per_instruction_of_a_wave\": %.3f, "
             "\"simd_instructions_per_cycle\": %.4f, \"lanes_per_cycle_per_simd\": %.2f, \"kernel_ms\": %.4f, \"ns_per_instruction_of_a_wave\": %.4f, "
             "\"cycle_counter_mhz_implied\": %.1f}",
             first ? "" : ",", kNames[KIND], waves_per_simd, instrs, cpi, waves_per_simd / cpi, 64.0 * waves_per_simd / cpi, ms,
             ms * 1e6 / instrs, (double)mx / (ms * 1e3));
    json += buf;
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    int dev = 0;
    CHK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, dev));
    const int n_cus = prop.multiProcessorCount;
    float* dout; unsigned long long* dcyc;
    CHK(hipMalloc(&dout, sizeof(float) * (size_t)n_cus * 512));
    CHK(hipMalloc(&dcyc, sizeof(unsigned long long) * (size_t)n_cus * 8));
    std::string tests;
    bool first = true;
    for (int w = 1; w <= 2; w++) {
        run<FMA32>(n_cus, w, dout, dcyc, tests, first); first = false;
        run<PKFMA32>(n_cus, w, dout, dcyc, tests, first);
        run<ADD32_DPP>(n_cus, w, dout, dcyc, tests, first);
        run<FMA64>(n_cus, w, dout, dcyc, tests, first);
        run<ADD32_CHAIN>(n_cus, w, dout, dcyc, tests, first);
        run<ADD32_DPP_CHAIN>(n_cus, w, dout, dcyc, tests, first);
        run<FMA64_CHAIN>(n_cus, w, dout, dcyc, tests, first);
        run<CVT_UBYTE>(n_cus, w, dout, dcyc, tests, first);
    }
    // the LDS-DMA stream: bytes known exactly; run this binary under rocprofv3 --pmc FETCH_SIZE to calibrate the counter
    const size_t chunks_per_wave = 4096;  // x 1 KiB
    const size_t bytes = (size_t)n_cus * 4 * chunks_per_wave * 1024;
    uint4* src; unsigned* sink;
    CHK(hipMalloc(&src, bytes)); CHK(hipMemset(src, 1, bytes)); CHK(hipMalloc(&sink, sizeof(unsigned) * n_cus));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(lds_dma_stream, dim3(n_cus), dim3(256), 0, 0, src, chunks_per_wave, sink);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(lds_dma_stream, dim3(n_cus), dim3(256), 0, 0, src, chunks_per_wave, sink);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"n_cus\": %d, \"clock_rate_khz\": %d,\n  \"tests\": [%s\n  ],\n"
           "  \"lds_dma_stream\": {\"kernel\": \"lds_dma_stream\", \"launches\": 2, \"bytes_per_launch\": %zu, \"kernel_ms\": %.4f, \"GBps\": %.1f}\n}\n",
           prop.name, prop.gcnArchName, n_cus, prop.clockRate, tests.c_str(), bytes, ms, bytes / (ms * 1e6));
    return 0;
}
