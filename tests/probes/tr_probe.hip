#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const unsigned short* in, unsigned short* out, const int* addr_elems) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  int a = addr_elems[threadIdx.x];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + a));
  for (int j = 0; j < 4; ++j) out[threadIdx.x*4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short h_in[4096], h_out[256]; int h_addr[64];
  for (int i = 0; i < 4096; ++i) h_in[i] = i;
  // experiment A: natural [4 rows][16 cols] blocks per 16-lane group, row stride 16 elements; lane L -> row L/4, cols (L%4)*4
  for (int l = 0; l < 64; ++l) { int L = l & 15, grp = l >> 4; h_addr[l] = grp*64 + (L/4)*16 + (L%4)*4; }
  unsigned short *d_in, *d_out; int* d_addr;
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out)); hipMalloc(&d_addr, sizeof(h_addr));
  for (int exp = 0; exp < 3; ++exp) {
    if (exp == 1) for (int l = 0; l < 64; ++l) { int L = l & 15, grp = l >> 4; h_addr[l] = grp*1000 + (L/4)*100 + (L%4)*4; }   // big row stride 100
    if (exp == 2) for (int l = 0; l < 64; ++l) { h_addr[l] = l*8; }   // each lane its own 8B-aligned chunk (lane*4 elems *2)
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice); hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, d_out, d_addr);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("exp %d\n", exp);
    for (int l = 0; l < 64; ++l) { printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h_addr[l], h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]); }
  }
  return 0;
}
