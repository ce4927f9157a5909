#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* off, short* out) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + off[threadIdx.x]));
    for (int k = 0; k < 4; ++k) out[threadIdx.x * 4 + k] = v[k];
}
int main() {
    int h_off[64]; short h_out[256];
    int *d_off; short* d_out;
    hipMalloc(&d_off, sizeof(h_off)); hipMalloc(&d_out, sizeof(h_out));
    for (int mode = 0; mode < 2; ++mode) {
        // mode 0: lane L points at row L (row stride 64 elements), column 0;  mode 1: lane L -> row (L & 15), column 4 * (L >> 4)
        for (int l = 0; l < 64; ++l) h_off[l] = mode == 0 ? l * 64 : (l & 15) * 64 + 4 * (l >> 4);
        hipMemcpy(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d_off, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("mode %d (element index = 64 * row + col)\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int k = 0; k < 4; ++k) printf(" (r%2d,c%2d)", h_out[l * 4 + k] / 64, h_out[l * 4 + k] % 64);
            printf("\n");
        }
    }
    return 0;
}
