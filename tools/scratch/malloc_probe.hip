// probe: cold costs of 64 host threads uploading table parts (persistent threads, fresh pageable buffers per round)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include <condition_variable>
#include <mutex>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const bool prewarm = argc > 1;
  hipSetDevice(0); hipFree(0);
  const size_t sz = 3u << 20;
  if (prewarm) {   // ONE thread: a big allocation and a big pageable copy before the workers start
    double a = now();
    void* d; hipMalloc(&d, 768u << 20);
    std::vector<char> h(16u << 20, 1);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipMemcpyAsync(d, h.data(), h.size(), hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
    hipFree(d);
    printf("main-thread prewarm (768 MB hipMalloc + 16 MB pageable H2D + free): %.1f ms\n", (now() - a) * 1e3);
  }
  const int T = 64, R = 4;
  std::vector<std::vector<double>> dt(R, std::vector<double>(T)), dm(R, std::vector<double>(T)), dc(R, std::vector<double>(T));
  std::atomic<int> round{-1}; std::atomic<int> done{0};
  std::vector<std::thread> th;
  std::vector<hipStream_t> streams(T);
  for (int w = 0; w < T; w++) th.emplace_back([&, w] {
    hipSetDevice(0);
    hipStreamCreateWithFlags(&streams[w], hipStreamNonBlocking);
    done++;
    for (int r = 0; r < R; r++) {
      while (round.load() < r) std::this_thread::yield();
      double s = now();
      std::vector<char> h(sz * 3, 1);   // fresh pageable memory, first touch here
      double a = now();
      void* d[3]; for (auto& x : d) hipMalloc(&x, sz);
      double m = now();
      for (int k = 0; k < 3; k++) hipMemcpyAsync(d[k], h.data() + k * sz, sz, hipMemcpyHostToDevice, streams[w]);
      hipStreamSynchronize(streams[w]);
      double e = now();
      dt[r][w] = a - s; dm[r][w] = m - a; dc[r][w] = e - m;
      for (auto& x : d) hipFree(x);
      done++;
    }
  });
  while (done.load() < T) std::this_thread::yield();
  for (int r = 0; r < R; r++) {
    done = 0; double a = now(); round = r;
    while (done.load() < T) std::this_thread::yield();
    double wall = now() - a, mt = 0, mm = 0, mc = 0;
    for (int w = 0; w < T; w++) { mt = std::max(mt, dt[r][w]); mm = std::max(mm, dm[r][w]); mc = std::max(mc, dc[r][w]); }
    printf("%s round %d: wall %.1f ms | slowest: host alloc+touch %.1f, 3 x hipMalloc %.1f, 3 x 3 MB pageable H2D + sync %.1f ms\n", prewarm ? "prewarmed" : "cold", r, wall * 1e3, mt * 1e3, mm * 1e3, mc * 1e3);
  }
  for (auto& t : th) t.join();
  return 0;
}
