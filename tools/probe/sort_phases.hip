// Where does a one-sweep pass of the depth sort spend its time? 4 views x P depth keys (floats in [3.5, 7), like an object
// seen from radius 5.2), sorted exactly as binning.hip does (iota values, drop form), timed per launch with HIP events and
// per PHASE with realtime stamps inside k_os_pass (GSR_OS_TRACE).
// build: hipcc -O3 --offload-arch=gfx950 -DGSR_OS_TRACE [-DGSR_OS_WINDOW=..] [-DPROBE_ITEMS=..] -I../../dreamscene_amd/csrc
//        -I../../include sort_phases.hip -o sort_phases
#define GSR_OS_TRACE 1
#include "radix_sort.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>
#ifndef PROBE_ITEMS
#define PROBE_ITEMS kOsItemsSmall
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const uint64_t P = argc > 1 ? strtoull(argv[1], 0, 10) : 500000;
  const int V = 4, reps = 30;
  const size_t kb = align256(P * 4), hb = sort_hist_bytes(P, kItemsSmall, PROBE_ITEMS);
  const size_t bstride = 4 * kb + hb + 256 + align256(kRadix * 4);
  char* slab; CK(hipMalloc(&slab, bstride * V));
  uint32_t *k0 = (uint32_t*)slab, *v0 = (uint32_t*)(slab + kb), *k1 = (uint32_t*)(slab + 2 * kb), *v1 = (uint32_t*)(slab + 3 * kb);
  uint32_t* hist = (uint32_t*)(slab + 4 * kb);
  uint64_t* ncomp = (uint64_t*)(slab + 4 * kb + hb);
  uint32_t* totals = (uint32_t*)(slab + 4 * kb + hb + 256);
  std::vector<uint32_t> h(P);
  std::vector<std::vector<uint32_t>> keys(V);
  srand(1);
  for (int v = 0; v < V; ++v) {
    for (uint64_t i = 0; i < P; ++i) {
      float z = 3.5f + 3.5f * (float)((double)rand() / RAND_MAX) + 1e-4f * (float)(rand() & 1023);
      if ((rand() & 63) == 0) { h[i] = 0xFFFFFFFFu; continue; }      // culled
      memcpy(&h[i], &z, 4);
    }
    keys[v] = h;
  }
  unsigned long long* trace; const size_t tw = (size_t)4 * 4 * 4096 * 8;
  CK(hipMalloc(&trace, tw * 8)); CK(hipMemset(trace, 0, tw * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_os_trace), &trace, sizeof(trace)));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float tot = 0;
  int where = 0;
  for (int r = 0; r < reps + 3; ++r) {
    for (int v = 0; v < V; ++v) CK(hipMemcpyAsync(slab + v * bstride, keys[v].data(), P * 4, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    where = radix_sort_u32<kItemsSmall, PROBE_ITEMS>(k0, v0, k1, v1, nullptr, P, 32, true, ncomp, hist, totals, st, V, bstride);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 3) tot += ms;
  }
  printf("P=%llu x %d views, items=%d window=%d: whole sort %.1f us (zero + hist + 4 passes)\n", (unsigned long long)P, V, (int)PROBE_ITEMS,
         (int)kOsWindow, tot / reps * 1e3);
  // correctness against std::stable_sort for view 0
  {
    uint64_t n; CK(hipMemcpy(&n, ncomp, 8, hipMemcpyDeviceToHost));
    std::vector<uint32_t> ok(n), ov(n);
    CK(hipMemcpy(ok.data(), where ? k1 : k0, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ov.data(), where ? v1 : v0, n * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> idx;
    for (uint64_t i = 0; i < P; ++i) if (keys[0][i] != 0xFFFFFFFFu) idx.push_back((uint32_t)i);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return keys[0][a] < keys[0][b]; });
    bool good = idx.size() == n;
    for (uint64_t i = 0; good && i < n; ++i) good = (ov[i] == idx[i]) && (ok[i] == keys[0][idx[i]]);
    printf("view 0: %llu survivors, %s\n", (unsigned long long)n, good ? "matches std::stable_sort" : "MISMATCH");
  }
  std::vector<unsigned long long> tr(tw);
  CK(hipMemcpy(tr.data(), trace, tw * 8, hipMemcpyDeviceToHost));
  const uint32_t ntile = os_tiles(P, PROBE_ITEMS);
  for (int pass = 0; pass < 4; ++pass) {
    double ph[5] = {0, 0, 0, 0, 0}; unsigned long long lo = ~0ull, hi = 0; int cnt = 0;
    double lb_by_pos[4] = {0, 0, 0, 0}; int lb_n[4] = {0, 0, 0, 0};
    for (int v = 0; v < V; ++v)
      for (uint32_t b = 0; b < ntile; ++b) {
        const unsigned long long* t = &tr[(((size_t)pass * 4 + v) * 4096 + b) * 8];
        if (!t[0] || !t[5]) continue;
        for (int k = 0; k < 5; ++k) ph[k] += (double)(t[k + 1] - t[k]);
        lo = std::min(lo, t[0]); hi = std::max(hi, t[5]); ++cnt;
        const int q = (int)std::min<unsigned long long>(3, t[7] * 4 / ntile);
        lb_by_pos[q] += (double)(t[3] - t[2]); ++lb_n[q];
      }
    if (!cnt) { printf("pass %d: identity copy (no stamps)\n", pass); continue; }
    printf("pass %d: %d tiles, span %.1f us | ticket+scan %.2f  load+rank %.2f  publish+look-back %.2f  lds sort %.2f  write %.2f us (mean per tile) | look-back by list quarter: %.2f %.2f %.2f %.2f\n",
           pass, cnt, (hi - lo) * 0.01, ph[0] / cnt * 0.01, ph[1] / cnt * 0.01, ph[2] / cnt * 0.01, ph[3] / cnt * 0.01, ph[4] / cnt * 0.01,
           lb_n[0] ? lb_by_pos[0] / lb_n[0] * 0.01 : 0, lb_n[1] ? lb_by_pos[1] / lb_n[1] * 0.01 : 0,
           lb_n[2] ? lb_by_pos[2] / lb_n[2] * 0.01 : 0, lb_n[3] ? lb_by_pos[3] / lb_n[3] * 0.01 : 0);
  }
  // pass 0, view 0: per-ticket timeline (us after the first stamp of the launch)
  {
    unsigned long long lo = ~0ull;
    std::vector<const unsigned long long*> by(ntile, nullptr);
    for (uint32_t b = 0; b < ntile; ++b) {
      const unsigned long long* t = &tr[(((size_t)0 * 4 + 0) * 4096 + b) * 8];
      if (!t[0] || !t[5]) continue;
      lo = std::min(lo, t[0]);
      if (t[7] < ntile) by[t[7]] = t;
    }
    printf("ticket: start  ranked  looked-back  end   (us since the launch's first stamp; pass 0, view 0)\n");
    for (uint32_t i = 0; i < ntile; i = i < 20 ? i + 1 : i + ntile / 16) {
      if (!by[i]) continue;
      printf("%6u: %6.2f %6.2f %6.2f %6.2f\n", i, (by[i][0] - lo) * 0.01, (by[i][2] - lo) * 0.01, (by[i][3] - lo) * 0.01, (by[i][5] - lo) * 0.01);
    }
  }
  return 0;
}
