// What does a raw buffer_load_dwordx4 return when it straddles num_records?  (gfx950 semantics probe)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const float* src, float* out, int n) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n * 4, 0x00020000);
    const int off = (n - 4 + (int)threadIdx.x) * 4;      // thread t starts t floats before the end minus 4: t=0 fully inside, t=1..3 straddle, t=4 fully out
    const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    out[threadIdx.x * 4 + 0] = __uint_as_float(v.x); out[threadIdx.x * 4 + 1] = __uint_as_float(v.y);
    out[threadIdx.x * 4 + 2] = __uint_as_float(v.z); out[threadIdx.x * 4 + 3] = __uint_as_float(v.w);
}
int main() {
    const int n = 64;
    float h[n + 8]; for (int i = 0; i < n + 8; ++i) h[i] = 100.f + i;
    float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 6>>>(d, o, n);
    float r[24]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int t = 0; t < 6; ++t) printf("start n-4+%d: %g %g %g %g\n", t, r[t*4], r[t*4+1], r[t*4+2], r[t*4+3]);
    return 0;
}
