// Microbenchmark: what does an all-to-all exchange of one row per workgroup cost INSIDE a kernel on MI355X (all workgroups resident), done
// with self-validating words {epoch : 32 | payload : 32} — agent-scope relaxed atomic stores / loads, no barrier object, no fence — the way
// rolo_peer_* exchanges LM sums between ranks? This prices a persistent LM kernel against one launch per trial.
//   hipcc --offload-arch=gfx950 -O3 -o grid_exchange grid_exchange.hip && ./grid_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void exch(unsigned long long* rows /* [2][nwg][words] */, int words, int trials, double* out, int* err, int work) {
  extern __shared__ unsigned xw[];
  const int nwg = gridDim.x, wg = blockIdx.x, t = threadIdx.x;
  double acc = 0;
  for (int tr = 1; tr <= trials; tr++) {
    const int par = tr & 1;
    // "the pass": some dependent arithmetic standing in for the body (work = 0: the exchange alone)
    double v = (double)(wg + tr);
    for (int k = 0; k < work; k++) v = v * 1.0000001 + 0.5;
    if (t < words) __hip_atomic_store(rows + ((size_t)par * nwg + wg) * words + t, ((unsigned long long)(unsigned)tr << 32) | (unsigned)(wg * 131 + t + (int)v % 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int total = nwg * words;
    const long long t0 = wall_clock64();
    for (int base = t; base < total; base += THREADS * 4) {
      unsigned long long x[4]; const unsigned long long* p[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int i = base + u * THREADS; p[u] = rows + (size_t)par * nwg * words + (i < total ? i : 0); x[u] = __hip_atomic_load(p[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = base + u * THREADS;
        if (i >= total) continue;
        while ((unsigned)(x[u] >> 32) != (unsigned)tr) {
          if (wall_clock64() - t0 > 200000000ll) { *err = 1; break; }   // 2 s at 100 MHz
          __builtin_amdgcn_s_sleep(1);
          x[u] = __hip_atomic_load(p[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        xw[i] = (unsigned)x[u];
      }
    }
    __syncthreads();
    if (t < words) { double s = 0; for (int r = 0; r < nwg; r++) s += (double)xw[r * words + t]; acc += s; }   // fixed order, every workgroup the same bits
    __syncthreads();
  }
  if (t == 0) out[wg] = acc;
}
int main() {
  int dev = 0; CHK(hipSetDevice(dev));
  for (int words : {24, 60}) for (int nwg : {64, 94, 128, 256}) for (int work : {0, 2000}) {
    unsigned long long* rows; double* out; int* err;
    CHK(hipMalloc(&rows, sizeof(unsigned long long) * 2 * nwg * words)); CHK(hipMemset(rows, 0, sizeof(unsigned long long) * 2 * nwg * words));
    CHK(hipMalloc(&out, sizeof(double) * nwg)); CHK(hipHostMalloc(&err, sizeof(int))); *err = 0;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int trials = 200; const size_t lds = sizeof(unsigned) * nwg * words;
    exch<512><<<nwg, 512, lds>>>(rows, words, 8, out, err, work); CHK(hipDeviceSynchronize());
    CHK(hipMemset(rows, 0, sizeof(unsigned long long) * 2 * nwg * words));
    CHK(hipEventRecord(e0)); exch<512><<<nwg, 512, lds>>>(rows, words, trials, out, err, work); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> h(nwg); CHK(hipMemcpy(h.data(), out, sizeof(double) * nwg, hipMemcpyDeviceToHost));
    bool same = true; for (int i = 1; i < nwg; i++) same &= h[i] == h[0];
    printf("words %2d workgroups %3d work %4d: %.2f us per trial (timeout %d, all workgroups agree %d)\n", words, nwg, work, 1e3 * ms / trials, *err, (int)same);
    CHK(hipFree(rows)); CHK(hipFree(out)); CHK(hipHostFree(err));
  }
  return 0;
}
