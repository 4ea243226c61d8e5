// Calibration for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 (MI355X_MICROARCH.md, HBM section):
// reads and writes a known number of bytes with the same access shape as nfc_demod_kernel's ring traffic
// (one dword per lane, 256-byte rows per wave) over a buffer far larger than L2 + Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void read_rows(const float *__restrict__ in, float *out, size_t n)
{
   size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   size_t stride = (size_t)gridDim.x * blockDim.x;
   float acc = 0;
   for (; i < n; i += stride)
      acc += in[i];
   if (acc == 12345.678f)
      out[0] = acc;
}

__global__ void write_rows(float *out, size_t n)
{
   size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   size_t stride = (size_t)gridDim.x * blockDim.x;
   for (; i < n; i += stride)
      out[i] = (float)i;
}

int main()
{
   const size_t n = (size_t)1 << 31; // 8 GiB of floats
   float *a = nullptr, *b = nullptr;
   if (hipMalloc(&a, n * 4) != hipSuccess || hipMalloc(&b, 4096) != hipSuccess)
      return 1;
   hipMemset(a, 0, n * 4);
   hipDeviceSynchronize();
   read_rows<<<4096, 64>>>(a, b, n);
   hipDeviceSynchronize();
   write_rows<<<4096, 64>>>(a, n);
   hipDeviceSynchronize();
   printf("calib bytes_read=%zu bytes_written=%zu\n", n * 4, n * 4);
   return 0;
}
