// Test shim (CPU): reads a CTVW0001 file with include/ctvio_window_io.hpp and writes it back -- the file a C++ caller of libctvio would load
// to run the same synthetic windows as bench.py (no GPU needed here: nothing is solved).
#include "../include/ctvio_window_io.hpp"
int main(int argc, char **argv) {
  if (argc != 3) return 2;
  std::vector<std::unique_ptr<ctvio::OwnedWindow>> ws;
  const std::string e = ctvio::load_windows(argv[1], ws);
  if (!e.empty()) { std::fprintf(stderr, "%s\n", e.c_str()); return 1; }
  std::vector<ctvio_window> flat;
  long long V = 0, M = 0;
  for (auto &o : ws) { flat.push_back(o->w); V += o->w.V; M += o->w.M; }
  std::printf("%zu windows, %lld visual blocks, %lld IMU samples, P0 = %d\n", ws.size(), V, M, ws.empty() ? 0 : 6 * ws[0]->w.K + 6 * ws[0]->w.F + 1);
  const std::string e2 = ctvio::save_windows(argv[2], flat.data(), (int32_t)flat.size());
  if (!e2.empty()) { std::fprintf(stderr, "%s\n", e2.c_str()); return 1; }
  return 0;
}
