// tr_probe.hip — what does ds_read_b64_tr_b16 return?  LDS holds u16 value = its own element index; every lane reads at a chosen address.
// Prints, per lane, the four 16-bit elements it received.   hipcc --offload-arch=gfx950 -O2 -o tr_probe tr_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const uint32_t* addr_bytes, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds + addr_bytes[threadIdx.x];
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x * 2] = v[0];
  out[threadIdx.x * 2 + 1] = v[1];
}

static void run(const char* name, const std::vector<uint32_t>& addr) {
  uint32_t *da, *dout;
  hipMalloc(&da, 64 * 4); hipMalloc(&dout, 128 * 4);
  hipMemcpy(da, addr.data(), 64 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, dout);
  std::vector<uint32_t> o(128);
  hipMemcpy(o.data(), dout, 128 * 4, hipMemcpyDeviceToHost);
  printf("== %s\n", name);
  for (int l = 0; l < 64; ++l)
    printf("lane %2d addr %4u -> elems %4u %4u %4u %4u\n", l, addr[l] / 2, o[2 * l] & 0xffff, o[2 * l] >> 16, o[2 * l + 1] & 0xffff, o[2 * l + 1] >> 16);
  hipFree(da); hipFree(dout);
}

int main() {
  std::vector<uint32_t> a(64);
  // pattern 1: lane l reads 8 bytes at l*8 (4 consecutive elements 4l..4l+3)
  for (int l = 0; l < 64; ++l) a[l] = l * 8;
  run("addr = lane*8 (elements 4l..4l+3)", a);
  // pattern 2: pixel-major image [pixel][64 ch] (128 B per pixel): within each 16-lane group, lane i -> pixel (i/4), channels 4*(i%4)..+3;
  // group g (l>>4) -> channel block 16*g
  for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; a[l] = ((i / 4) * 64 + 16 * g + 4 * (i % 4)) * 2; }
  run("pixel-major [p][64c]: lane i of group g -> (pixel i/4, channels 16g + 4(i%4)..)", a);
  // pattern 3: lane i -> pixel (i%4), channels 4*(i/4)
  for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; a[l] = ((i % 4) * 64 + 16 * g + 4 * (i / 4)) * 2; }
  run("pixel-major [p][64c]: lane i of group g -> (pixel i%4, channels 16g + 4(i/4)..)", a);
  return 0;
}
