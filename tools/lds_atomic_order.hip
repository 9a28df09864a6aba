// Does a returning LDS atomic serve the lanes of ONE wave instruction in lane order when several lanes hit the same address?
// (The split blend could append to per-pixel lists with one ds_add_rtn_u32 instead of its mask + v_mbcnt ranking if so.)
// Random address patterns with 1 .. 64 lanes per address, random active masks; for every address the returned values must increase
// with the lane number.  build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_order.hip -o tools/_build/lds_atomic_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k(const unsigned *__restrict__ addr, const unsigned long long *__restrict__ active, unsigned *__restrict__ ret, int rounds)
{
    __shared__ unsigned cnt[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = 0; r < rounds; ++r) {
        cnt[wave][lane] = 0;
        const size_t o = ((size_t)(blockIdx.x * rounds + r) * 4 + wave) * 64 + lane;
        const unsigned a = addr[o] & 63u;
        const bool on = (active[o / 64] >> lane) & 1ull;
        unsigned v = 0xFFFFFFFFu;
        if (on) v = __hip_atomic_fetch_add(&cnt[wave][a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ret[o] = v;
    }
}

int main()
{
    const int wgs = 2048, rounds = 16;
    const size_t n = (size_t)wgs * rounds * 4 * 64;
    std::vector<unsigned> addr(n), ret(n);
    std::vector<unsigned long long> act(n / 64);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t w = 0; w < n / 64; ++w) {
        const int spread = 1 << (rnd() % 7);   // 1 .. 64 distinct addresses
        for (int l = 0; l < 64; ++l) addr[w * 64 + l] = (unsigned)((rnd() % spread) * (64 / spread) + (rnd() % 3 == 0 ? 0 : 0));
        act[w] = (rnd() % 4 == 0) ? ~0ull : rnd() | rnd();
    }
    unsigned *da, *dr; unsigned long long *dm;
    hipMalloc(&da, n * 4); hipMalloc(&dr, n * 4); hipMalloc(&dm, n / 64 * 8);
    hipMemcpy(da, addr.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dm, act.data(), n / 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, da, dm, dr, rounds);
    hipDeviceSynchronize();
    hipMemcpy(ret.data(), dr, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, conflicts = 0;
    for (size_t w = 0; w < n / 64; ++w) {
        unsigned next[64] = {0};
        for (int l = 0; l < 64; ++l) {
            if (!((act[w] >> l) & 1ull)) continue;
            const unsigned a = addr[w * 64 + l] & 63u;
            if (next[a]) ++conflicts;
            if (ret[w * 64 + l] != next[a]) ++bad;
            ++next[a];
        }
    }
    printf("lds_atomic_order: %zu wave instructions, %zu conflicting lanes, %zu lanes out of lane order -> %s\n", n / 64, conflicts, bad,
           bad ? "NOT in lane order" : "lane order held in every case");
    return 0;
}
