// Where does global_load_lds_dwordx3 / dwordx4 put each lane's bytes?  One wave copies
// 64 x size bytes from a tagged global array (lane l reads from src + l * 8 dwords);
// the LDS image is dumped.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define KERNEL(SIZE)                                                                                   \
    __global__ void k##SIZE(const uint32_t* __restrict__ src, uint32_t* __restrict__ out)              \
    {                                                                                                  \
        extern __shared__ __align__(16) uint32_t lds[];                                                \
        const int lane = threadIdx.x;                                                                  \
        for (int i = lane; i < 512; i += 64) lds[i] = 0xdeadbeef;                                      \
        __syncthreads();                                                                               \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 8), \
                                         (__attribute__((address_space(3))) void*)(lds), SIZE, 0, 0);  \
        __syncthreads();                                                                               \
        for (int i = lane; i < 512; i += 64) out[i] = lds[i];                                          \
    }
KERNEL(4)
KERNEL(12)
KERNEL(16)

int main()
{
    std::vector<uint32_t> h(64 * 8);
    for (int l = 0; l < 64; l++) for (int d = 0; d < 8; d++) h[l * 8 + d] = (l << 8) | d;
    uint32_t *s, *o;
    hipMalloc(&s, h.size() * 4); hipMalloc(&o, 512 * 4);
    hipMemcpy(s, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> r(512);
    for (int size : {4, 12, 16}) {
        if (size == 4) k4<<<1, 64, 2048>>>(s, o);
        if (size == 12) k12<<<1, 64, 2048>>>(s, o);
        if (size == 16) k16<<<1, 64, 2048>>>(s, o);
        hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
        printf("size %d: first 24 dwords (lane<<8|dword):", size);
        for (int i = 0; i < 24; i++) printf(" %x", r[i]);
        printf("\n   dwords 186..200:");
        for (int i = 186; i < 200; i++) printf(" %x", r[i]);
        printf("\n");
    }
    return 0;
}
