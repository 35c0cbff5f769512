// Micro-benchmark: issue cost of the VALU / SALU instruction kinds the gather kernels are made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// One workgroup per CU-slot, W waves per SIMD; each wave runs REP x 64 independent instructions of one kind;
// cycles per instruction per SIMD = elapsed shader cycles / (REP * 64 * W).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256
#define CHAIN8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int KIND>
__global__ void k(unsigned* out, unsigned long long* cyc) {
  unsigned v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 7 + i;
  unsigned s = blockIdx.x;
  unsigned long long mask[4] = {0, 0, 0, 0};
  __shared__ unsigned lds[4096];
  const unsigned laddr = (threadIdx.x & 1023) * 16;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  u4 wide[2] = {};
  if (threadIdx.x > 100000) lds[threadIdx.x] = 1;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (KIND == 0) {
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 1) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 2) {
#define OP(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 3) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 4) {
#define OP(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 5) {
#define OP(i) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 6) {
#define OP(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 7) {
#define OP(i) asm volatile("v_readlane_b32 %0, %1, 3" : "+s"(s) : "v"(v[i]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 8) {
#define OP(i) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 9) {
#define OP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(v[i]), "v"(v[(i + 1) & 7]) : "vcc");
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 10) {
#define OP(i) asm volatile("v_cvt_f32_i32 %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 11) {
#define OP(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 12) {
#define OP(i) asm volatile("v_lshlrev_b32 %0, 3, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 13) {
#define OP(i) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 14) {
#define OP(i) asm volatile("v_add_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 15) {
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 16) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(unsigned long long*)&v[(i & 3) * 2]) : "v"(*(unsigned long long*)&v[((i + 1) & 3) * 2]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 17) {
#define OP(i) asm volatile("v_min_i32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 18) {
#define OP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[(i + 1) & 7]) : "vcc");
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 19) {
#define OP(i) asm volatile("v_cmp_lt_u32 %2, %0, %1\n\tv_cndmask_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "s"(mask[i & 3]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 20) {
#define OP(i) asm volatile("v_floor_f32 %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 21) {
#define OP(i) asm volatile("v_cvt_i32_f32 %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 22) {
#define OP(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 23) {
#define OP(i) asm volatile("v_rcp_f32 %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 24) {
#define OP(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 25) {
#define OP(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 26) {
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 27) {
#define OP(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 28) {
#define OP(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 29) {
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 30) {
#define OP(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(4)" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 31) {
#define OP(i) asm volatile("ds_write_b32 %0, %1" : : "v"(laddr), "v"(v[i]) : "memory");
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 32) {
#define OP(i) asm volatile("ds_write_b16_d16_hi %0, %1" : : "v"(laddr), "v"(v[i]) : "memory");
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 33) {
#define OP(i) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(6)" : "=v"(wide[i & 1]) : "v"(laddr) : "memory");
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 34) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(unsigned long long*)&v[(i & 3) * 2]) : "v"(*(unsigned long long*)&v[((i + 1) & 3) * 2]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 35) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 36) {
#define OP(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 37) {
#define OP(i) asm volatile("v_exp_f32 %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 38) {
#define OP(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 39) {
#define OP(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 40) {
#define OP(i) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 41) {
#define OP(i) asm volatile("v_fract_f32 %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 42) {
#define OP(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      } else if constexpr (KIND == 43) {
#define OP(i) asm volatile("v_bfe_u32 %0, %1, 3, 5" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        CHAIN8(OP)
#undef OP
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  unsigned acc = s + wide[0][0] + wide[1][1] + (unsigned)mask[0];
  for (int i = 0; i < 8; ++i) acc += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
  unsigned* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4);
  hipMalloc(&cyc, 1024 * 8);
  for (int w : {1, 2, 4}) {          // waves per SIMD (workgroup = 4 SIMDs x w waves), one workgroup per CU
    const int threads = 256 * w;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto c : h) avg += c;
    avg /= 256;
    // s_memtime counts at 100 MHz on gfx9 (constant clock); wall time is the reliable figure: ns per instruction per SIMD
    const double n_inst = double(REP) * 64 * w;
    printf("%-16s waves/SIMD %d: wall %.1f us -> %.3f ns per instr per SIMD (%.2f cyc @2.4GHz); memtime ticks/instr %.3f\n", name, w, ms * 1e3,
           ms * 1e6 / n_inst, ms * 1e6 / n_inst * 2.4, avg / n_inst);
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("v_add_u32");
  run<1>("v_fma_f32");
  run<15>("v_mul_f32");
  run<16>("v_pk_fma_f32");
  run<2>("v_perm_b32");
  run<3>("v_cndmask");
  run<4>("v_mul_u32_u24");
  run<11>("v_mad_u32_u24");
  run<5>("v_pk_min_i16");
  run<17>("v_min_i32");
  run<6>("v_mov_dpp");
  run<14>("v_add_u32_dpp");
  run<7>("v_readlane");
  run<8>("s_add_u32");
  run<9>("v_cmp");
  run<10>("v_cvt_f32_i32");
  run<12>("v_lshlrev");
  run<13>("v_and_or");
  run<18>("v_cmp+cndmask_vcc");
  run<19>("v_cmp+cndmask_sgpr");
  run<20>("v_floor_f32");
  run<21>("v_cvt_i32_f32");
  run<22>("v_permlane32_swap");
  run<23>("v_rcp_f32");
  run<24>("v_lshl_or_b32");
  run<25>("v_add3_u32");
  run<26>("v_mov_b32");
  run<27>("v_sub_f32");
  run<28>("v_and_b32");
  run<29>("v_mul_lo_u32");
  run<30>("ds_bpermute");
  run<31>("ds_write_b32");
  run<32>("ds_write_b16_hi");
  run<33>("ds_read_b128");
  run<34>("v_pk_mul_f32");
  run<35>("v_add_f32");
  run<36>("v_sub_u32");
  run<37>("v_exp_f32");
  run<38>("v_max_f32");
  run<39>("v_cvt_pk_bf16_f32");
  run<40>("v_med3_i32");
  run<41>("v_fract_f32");
  run<42>("v_xor_b32");
  run<43>("v_bfe_u32");
  return 0;
}
