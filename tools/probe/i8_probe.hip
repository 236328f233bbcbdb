#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* in, unsigned* out, int* mf) {
  int l = threadIdx.x;
  // cvt_pk_u8_f32 rounding
  if (l < 16) out[l] = __builtin_amdgcn_cvt_pk_u8_f32(in[l], 0, 0u);
  // i8 mfma layout probe: A[row][k] = (row==r0 && k==k0), B[k][col] = (k==k0 && col==c0) -> C[r0][c0]=1
  // lane l holds A[row l&15][k = 16*(l>>4)+j]; encode A[row][k] = row (as i8) for k== (16*(l>>4)+j), B[k][col] = (k == kk)
  for (int kk = 0; kk < 64; kk += 21) {
    i4 a, b, c = {0,0,0,0};
    unsigned char ab[16], bb[16];
    for (int j = 0; j < 16; ++j) { int k = 16*(l>>4)+j; ab[j] = (unsigned char)((l&15) + 1 + (k==kk ? 100 : 0) - (k==kk?100:0)); ab[j] = (k==kk) ? (unsigned char)((l&15)+1) : 0; bb[j] = (k==kk) ? (unsigned char)((l&15)+3) : 0; }
    __builtin_memcpy(&a, ab, 16); __builtin_memcpy(&b, bb, 16);
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) mf[(kk/21)*256 + l*4 + r] = c[r];
  }
}
int main() {
  float h[16] = {0.5f, 1.5f, 2.5f, 0.49f, 0.51f, 254.5f, 255.5f, 300.f, -0.4f, -3.f, 3.5f, 127.5f, 128.5f, 2.4999f, 1.0f, 254.49f};
  float* d; unsigned* o; int* mf;
  hipMalloc(&d, 64); hipMalloc(&o, 64); hipMalloc(&mf, 4*256*4);
  hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
  probe<<<1,64>>>(d, o, mf);
  unsigned ho[16]; int hm[4*256];
  hipMemcpy(ho, o, 64, hipMemcpyDeviceToHost); hipMemcpy(hm, mf, sizeof(hm), hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i) printf("%g->%u ", h[i], ho[i] & 255);
  printf("\n");
  for (int t = 0; t < 4; ++t) {
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { int row = 4*(l>>4)+r, col = l&15; if (hm[t*256+l*4+r] != (row+1)*(col+3)) ok = 0; }
    printf("mfma i8 kk=%d layout %s (c[lane0] = %d %d %d %d)\n", t*21, ok ? "OK" : "MISMATCH", hm[t*256], hm[t*256+1], hm[t*256+2], hm[t*256+3]);
  }
  return 0;
}
