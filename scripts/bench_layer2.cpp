// bench_layer2.cpp -- the operator boundary (layer 2, include/crgpu.h: crgpu_hj_*) driven the way a cgo shell would:
// host coldata batches of `bs` rows in, caller-owned `bs`-row host batches out, no interpreter in the loop
// (scripts/bench_layer2.py measures the same calls through ctypes, where the ~1 us per call of the binding is a third of
// the time).  INNER join, int64 key + int64 payload on both sides, every probe row matches; the output is checked by a
// checksum.  Also prints the host's single-thread memcpy rate over buffers of the same size: every input byte is
// copied once into the staging area and every output byte once out of it, so that rate bounds the boundary.
//
// Build: g++ -O2 -std=c++17 -I include scripts/bench_layer2.cpp -o scripts/_build/bench_layer2 -L cockroach_amd -lcrgpu
//        -Wl,-rpath,'$ORIGIN/../../cockroach_amd'      (scripts/Makefile)
// Usage: bench_layer2 [lg_build=22] [lg_probe=24] [batch_rows=1024] [lookahead_batches=1024] [borrow_output=0]
//        borrow_output=1: CRGPU_OUT_BORROW, the output values are read where the library's transfer put them
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "crgpu.h"

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CHECK(x)                                                                         \
  do {                                                                                   \
    crgpu_status _s = (x);                                                               \
    if (_s != CRGPU_OK) {                                                                \
      std::fprintf(stderr, "%s failed (%d): %s\n", #x, (int)_s, crgpu_last_error());    \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

struct OutBatch {
  std::vector<int64_t> a, b;
  std::vector<uint8_t> na, nb;
  crgpu_out_col cols[2];
  crgpu_out_batch ob;
  explicit OutBatch(int64_t cap, bool borrow) : a(cap), b(cap), na((cap + 7) / 8), nb((cap + 7) / 8) {
    std::memset(cols, 0, sizeof(cols));
    cols[0].type = CRGPU_INT64, cols[0].data = a.data(), cols[0].nulls = na.data();
    cols[1].type = CRGPU_INT64, cols[1].data = b.data(), cols[1].nulls = nb.data();
    ob.capacity = cap, ob.length = 0, ob.ncols = 2, ob.flags = borrow ? CRGPU_OUT_BORROW : 0, ob.cols = cols;
  }
};

int main(int argc, char **argv) {
  const int lg_b = argc > 1 ? std::atoi(argv[1]) : 22, lg_p = argc > 2 ? std::atoi(argv[2]) : 24;
  const int64_t bs = argc > 3 ? std::atoll(argv[3]) : 1024, lookahead = argc > 4 ? std::atoll(argv[4]) : 1024;
  const bool borrow = argc > 5 && std::atoi(argv[5]) != 0;
  const int64_t nb = int64_t(1) << lg_b, np = int64_t(1) << lg_p;
  CHECK(crgpu_init(0));
  std::mt19937_64 rng(42);
  std::vector<int64_t> bk(nb), bv(nb), pk(np), pw(np);
  for (int64_t i = 0; i < nb; i++) bk[i] = i;
  for (int64_t i = nb - 1; i > 0; i--) std::swap(bk[i], bk[rng() % (i + 1)]);
  for (int64_t i = 0; i < nb; i++) bv[i] = bk[i] * 3 + 1;
  uint64_t want = 0;
  for (int64_t i = 0; i < np; i++) {
    pk[i] = static_cast<int64_t>(rng() % nb);
    pw[i] = i;
    want += static_cast<uint64_t>(pw[i]) * 31u + static_cast<uint64_t>(pk[i] * 3 + 1);
  }
  const int32_t types[2] = {CRGPU_INT64, CRGPU_INT64}, eq[1] = {0}, outc[2] = {1, 3};
  crgpu_join_spec spec;
  std::memset(&spec, 0, sizeof(spec));
  spec.join_type = CRGPU_INNER;
  spec.right_eq_cols_are_key = 1;
  spec.n_left_cols = 2, spec.n_right_cols = 2;
  spec.left_types = types, spec.right_types = types;
  spec.n_eq = 1, spec.left_eq = eq, spec.right_eq = eq;
  spec.n_out_cols = 2, spec.out_cols = outc;
  crgpu_hj *h = nullptr;
  CHECK(crgpu_hj_new(&spec, &h));
  OutBatch out(bs, borrow);
  double build_s = 0, probe_s = 0, append_s = 0, drain_s = 0;
  for (int rep = 0; rep < 3; rep++) {  // the first pass warms the allocations
    if (rep) CHECK(crgpu_hj_reset(h));
    uint64_t got = 0;
    int64_t rows = 0;
    auto drain = [&]() -> int {
      for (;;) {
        CHECK(crgpu_hj_next_output(h, &out.ob));
        if (out.ob.length == 0) return 0;
        const int64_t *oa = static_cast<const int64_t *>(out.cols[0].data), *obv = static_cast<const int64_t *>(out.cols[1].data);
        for (int64_t i = 0; i < out.ob.length; i++) got += static_cast<uint64_t>(oa[i]) * 31u + static_cast<uint64_t>(obv[i]);
        rows += out.ob.length;
      }
    };
    const double t0 = now();
    for (int64_t s = 0; s < nb; s += bs) {
      crgpu_col c[2];
      std::memset(c, 0, sizeof(c));
      c[0].type = CRGPU_INT64, c[0].data = bk.data() + s;
      c[1].type = CRGPU_INT64, c[1].data = bv.data() + s;
      crgpu_batch b{std::min(bs, nb - s), nullptr, 2, 0, c};
      CHECK(crgpu_hj_build_append(h, &b));
    }
    CHECK(crgpu_hj_build_finish(h));
    const double t1 = now();
    int64_t nbatch = 0;
    double in_append = 0, in_drain = 0;
    for (int64_t s = 0; s < np; s += bs) {
      crgpu_col c[2];
      std::memset(c, 0, sizeof(c));
      c[0].type = CRGPU_INT64, c[0].data = pk.data() + s;
      c[1].type = CRGPU_INT64, c[1].data = pw.data() + s;
      crgpu_batch b{std::min(bs, np - s), nullptr, 2, 0, c};
      const double a0 = now();
      CHECK(crgpu_hj_probe_append(h, &b));
      const double a1 = now();
      in_append += a1 - a0;
      if (++nbatch % lookahead == 0) {  // the shell reads ahead
        if (drain()) return 1;
        in_drain += now() - a1;
      }
    }
    const double f0 = now();
    CHECK(crgpu_hj_probe_finish(h));
    if (drain()) return 1;
    const double t2 = now();
    in_drain += t2 - f0;
    append_s = in_append, drain_s = in_drain;
    if (rows != np || got != want) {
      std::fprintf(stderr, "result check failed: %lld rows (want %lld), checksum %llu (want %llu)\n", (long long)rows,
                   (long long)np, (unsigned long long)got, (unsigned long long)want);
      return 1;
    }
    build_s = t1 - t0, probe_s = t2 - t1;
  }
  crgpu_hj_free(h);
  // the host's memcpy rate in pieces of one batch column (what staging does), single thread
  std::vector<int64_t> dst(np);
  double best = 0;
  for (int r = 0; r < 3; r++) {
    const double m0 = now();
    for (int64_t s = 0; s < np; s += bs) std::memcpy(dst.data() + s, pw.data() + s, static_cast<size_t>(std::min(bs, np - s)) * 8);
    const double gbps = 8.0 * np / (now() - m0) / 1e9;
    if (gbps > best) best = gbps;
  }
  std::printf("{\"case\": \"layer 2 (host coldata batches of %lld rows in and out, PCIe included, C driver): INNER join 2^%d build x 2^%d probe, int64 key + int64 payload\", "
              "\"lookahead_batches\": %lld, \"borrow_output\": %d, \"rows_per_s\": %.4g, \"build_rows_per_s\": %.4g, \"probe_rows_per_s\": %.4g, "
              "\"host_GBps_in_plus_out\": %.3f, \"build_s\": %.5f, \"probe_s\": %.5f, \"probe_append_s\": %.5f, \"probe_drain_s\": %.5f, \"host_memcpy_GBps_1_thread\": %.2f, \"result_check\": \"ok\"}\n",
              (long long)bs, lg_b, lg_p, (long long)lookahead, borrow ? 1 : 0, (nb + np) / (build_s + probe_s), nb / build_s, np / probe_s,
              (16.0 * nb + 32.0 * np) / (build_s + probe_s) / 1e9, build_s, probe_s, append_s, drain_s, best);
  return 0;
}
