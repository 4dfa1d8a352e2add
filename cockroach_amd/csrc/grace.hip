// grace.hip -- Grace-partitioned (external) hash join and hash aggregation: HBM -> host memory -> disk
// (SURVEY.md 8(f) N4).
//
// Replaces colexecdisk.hashBasedPartitioner (pkg/sql/colexec/colexecdisk/hash_based_partitioner.go:143-717) with the
// hash joiner (external_hash_joiner.go:62-143) or the hash aggregator (external_hash_aggregator.go:32-95) as its
// in-memory "main" operator -- what the disk spiller (disk_spiller.go:81-200) switches to once the in-memory operator
// exceeds its budget.  The state machine is the reference's:
//   hbpInitialPartitioning   every input batch is distributed over num_buckets partitions by
//                            TupleHashDistributor(DefaultInitHashValue + 1) on the equality / grouping columns
//                            (crgpu_dev_partition: the reference's hash, :367-371) and the partitions' rows leave HBM
//                            -- first to host memory, beyond host_limit to files under spill_dir
//                            (colcontainer.PartitionedDiskQueue) --; a partition's size is that of its LAST input's
//                            rows (the build side, :413-419)
//   hbpProcessNewPartitionUsingMain / hbpProcessingUsingMain
//                            a partition whose size fits mem_limit comes back to HBM and runs through the in-memory
//                            operator (crgpu_dev_join_* / crgpu_dev_agg_*), Reset between partitions (:571-586)
//   hbpRecursivePartitioning a partition that does not fit is redistributed with the next hash seed (:482-560);
//                            children that shrank by less than 5 % (hbpRecursivePartitioningSizeDecreaseThreshold,
//                            rows that are not distinct on the hash columns) go to the fallback list
//   fallback                 the reference sorts and merge-joins / sort-aggregates such a partition from disk.  Here
//                            mem_limit is a BUDGET inside 288 GB of HBM, not the size of the memory: a fallback
//                            partition runs through the same in-memory operator with the budget lifted, and only a
//                            real device allocation failure (CRGPU_ERR_OOM) hands the query back to the CPU operator.
// Equal keys meet in one partition, so the concatenation of the partitions' outputs is the operator's output for
// every join type and for GROUP BY (groups of different partitions are disjoint).
//
// Host orchestration over the existing device operators; the only data movement it adds is the spill itself
// (D2H / H2D copies, PCIe-bound).
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <unistd.h>

#include "common.h"

namespace crgpu {

static constexpr double kSizeDecreaseThreshold = 0.05;          // hash_based_partitioner.go:100
static constexpr int64_t kMinimalMaxPartitionSize = 64 << 10;   // hbpMinimalMaxPartitionSizeForMain (:108)

struct SpillCol {
  std::vector<uint8_t> data;      // fixed-width values or BYTES payload
  std::vector<uint32_t> offsets;  // BYTES: n + 1, starting at 0
  std::vector<uint8_t> present;   // one byte per row; empty: no NULLs
};
struct SpillChunk {
  int64_t n = 0;
  int64_t bytes = 0;
  std::vector<SpillCol> cols;
  std::string file;  // non-empty: the columns live in this file (disk tier)
};
struct PartInfo {
  std::vector<SpillChunk> chunks[2];
  int64_t rows[2] = {0, 0};
  int64_t mem_size = 0;                  // bytes of the last input's rows
  int64_t parent_mem = INT64_MAX;
};

struct LoadedInput {
  std::vector<DevBuf> data, valid, offsets;
  std::vector<crgpu_dcol> cols;
  int64_t n = 0;
};

}  // namespace crgpu

using namespace crgpu;

struct crgpu_grace : crgpu::Handle {
  int kind = 0;  // 0 join, 1 aggregation
  int ninputs = 1;
  std::vector<int> types[2], hash_cols[2];
  crgpu_join *join = nullptr;
  crgpu_agg *agg = nullptr;
  int join_type = 0;
  int64_t mem_limit = 0, host_limit = INT64_MAX;
  int num_buckets = 16, max_repartitions = 64;
  std::string spill_dir;
  std::map<int, PartInfo> main_parts, fallback_parts;
  std::vector<int> fallback_order;
  int part_offset = 0;
  uint64_t seed = 2;  // colexechash.DefaultInitHashValue + 1 (hash_based_partitioner.go:367-371)
  std::map<uint64_t, crgpu_partitioner *> partitioners[2];
  bool partitioning = true, finished = false, emit_right_pending = false;
  LoadedInput loaded[2];
  int64_t host_bytes = 0;
  uint64_t file_seq = 0;
  crgpu_grace_stats stats{};
  ~crgpu_grace() {
    for (int i = 0; i < 2; i++)
      for (auto &kv : partitioners[i]) crgpu_dev_partitioner_destroy(kv.second);
    if (join != nullptr) crgpu_dev_join_destroy(join);
    if (agg != nullptr) crgpu_dev_agg_destroy(agg);
    for (auto *m : {&main_parts, &fallback_parts})
      for (auto &kv : *m)
        for (int i = 0; i < 2; i++)
          for (auto &c : kv.second.chunks[i])
            if (!c.file.empty()) std::remove(c.file.c_str());
  }
};

// ---------------------------------------------------------------------- spill store
static crgpu_status chunk_to_disk(crgpu_grace *g, SpillChunk &c) {
  char name[64];
  std::snprintf(name, sizeof(name), "/crgpu_spill_%d_%p_%llu.bin", static_cast<int>(getpid()), static_cast<void *>(g),
                static_cast<unsigned long long>(g->file_seq++));
  const std::string path = g->spill_dir + name;
  FILE *f = std::fopen(path.c_str(), "wb");
  if (f == nullptr) return fail(CRGPU_ERR_OOM, "crgpu_grace: cannot create a spill file under spill_dir");
  bool ok = true;
  for (const SpillCol &col : c.cols) {
    const uint64_t sz[3] = {col.data.size(), col.offsets.size(), col.present.size()};
    ok = ok && std::fwrite(sz, 8, 3, f) == 3;
    if (sz[0]) ok = ok && std::fwrite(col.data.data(), 1, sz[0], f) == sz[0];
    if (sz[1]) ok = ok && std::fwrite(col.offsets.data(), 4, sz[1], f) == sz[1];
    if (sz[2]) ok = ok && std::fwrite(col.present.data(), 1, sz[2], f) == sz[2];
  }
  ok = std::fclose(f) == 0 && ok;
  if (!ok) {
    std::remove(path.c_str());
    return fail(CRGPU_ERR_OOM, "crgpu_grace: writing a spill file failed (disk full?)");
  }
  c.file = path;
  g->stats.bytes_spilled_disk += c.bytes;
  for (SpillCol &col : c.cols) {
    std::vector<uint8_t>().swap(col.data);
    std::vector<uint32_t>().swap(col.offsets);
    std::vector<uint8_t>().swap(col.present);
  }
  return CRGPU_OK;
}

// the chunk's columns in host memory (read back from its file if that is where they are); the file is removed
static crgpu_status chunk_fetch(crgpu_grace *g, SpillChunk &c) {
  if (c.file.empty()) {
    g->host_bytes -= c.bytes;
    return CRGPU_OK;
  }
  FILE *f = std::fopen(c.file.c_str(), "rb");
  if (f == nullptr) return fail(CRGPU_ERR_HIP, "crgpu_grace: a spill file disappeared");
  bool ok = true;
  for (SpillCol &col : c.cols) {
    uint64_t sz[3] = {0, 0, 0};
    ok = ok && std::fread(sz, 8, 3, f) == 3;
    if (!ok) break;
    col.data.resize(sz[0]);
    col.offsets.resize(sz[1]);
    col.present.resize(sz[2]);
    if (sz[0]) ok = ok && std::fread(col.data.data(), 1, sz[0], f) == sz[0];
    if (sz[1]) ok = ok && std::fread(col.offsets.data(), 4, sz[1], f) == sz[1];
    if (sz[2]) ok = ok && std::fread(col.present.data(), 1, sz[2], f) == sz[2];
  }
  std::fclose(f);
  std::remove(c.file.c_str());
  c.file.clear();
  if (!ok) return fail(CRGPU_ERR_HIP, "crgpu_grace: reading a spill file failed");
  return CRGPU_OK;
}

// Distribute `n` device rows of input `input` over num_buckets partitions with the current seed and move every
// partition's rows out of HBM (hashBasedPartitioner.partitionBatch, :377-422).
static crgpu_status partition_and_spill(crgpu_grace *g, int input, const crgpu_dcol *cols, int64_t n, int64_t parent_mem) {
  if (n == 0) return CRGPU_OK;
  const int nc = static_cast<int>(g->types[input].size());
  // one partitioner per input: a repartition bumps the seed, and the handle of the previous seed (chunk-sized
  // destination / histogram / output buffers) would otherwise stay until the operator is destroyed -- up to 2 x
  // max_repartitions of them inside the operator whose job is to stay within mem_limit
  auto &held = g->partitioners[input];
  for (auto it = held.begin(); it != held.end();) {
    if (it->first != g->seed) {
      crgpu_dev_partitioner_destroy(it->second);
      it = held.erase(it);
    } else {
      ++it;
    }
  }
  crgpu_partitioner *&p = held[g->seed];
  if (p == nullptr) {
    CRGPU_TRY(crgpu_dev_partitioner_create(g->types[input].data(), nc, g->hash_cols[input].data(),
                                           static_cast<int32_t>(g->hash_cols[input].size()), g->seed, g->num_buckets, &p));
  }
  std::vector<int64_t> counts(g->num_buckets, 0);
  crgpu_dev_result res;
  CRGPU_TRY(crgpu_dev_partition(p, cols, n, &res, counts.data()));
  // one download per column, sliced per partition on the host
  std::vector<SpillCol> whole(nc);
  for (int c = 0; c < nc; c++) {
    const crgpu_dcol &d = res.cols[c];
    SpillCol &w = whole[c];
    if (d.type == CRGPU_BYTES) {
      w.offsets.resize(static_cast<size_t>(n) + 1);
      CRGPU_HIP(hipMemcpy(w.offsets.data(), d.offsets, (static_cast<size_t>(n) + 1) * 4, hipMemcpyDeviceToHost));
      w.data.resize(w.offsets[n]);
      if (w.offsets[n]) CRGPU_HIP(hipMemcpy(w.data.data(), d.data, w.offsets[n], hipMemcpyDeviceToHost));
    } else {
      w.data.resize(static_cast<size_t>(n) * type_width(d.type));
      CRGPU_HIP(hipMemcpy(w.data.data(), d.data, w.data.size(), hipMemcpyDeviceToHost));
    }
    if (d.valid != nullptr) {
      std::vector<uint8_t> bm(static_cast<size_t>((n + 7) / 8));
      CRGPU_HIP(hipMemcpy(bm.data(), d.valid, bm.size(), hipMemcpyDeviceToHost));
      w.present.resize(static_cast<size_t>(n));
      for (int64_t i = 0; i < n; i++) w.present[i] = (bm[i >> 3] >> (i & 7)) & 1;
    }
  }
  int64_t start = 0;
  for (int q = 0; q < g->num_buckets; q++) {
    const int64_t cnt = counts[q];
    if (cnt == 0) continue;
    SpillChunk ch;
    ch.n = cnt;
    ch.cols.resize(nc);
    for (int c = 0; c < nc; c++) {
      const SpillCol &w = whole[c];
      SpillCol &s = ch.cols[c];
      if (g->types[input][c] == CRGPU_BYTES) {
        const uint32_t b0 = w.offsets[start], b1 = w.offsets[start + cnt];
        s.offsets.resize(static_cast<size_t>(cnt) + 1);
        for (int64_t i = 0; i <= cnt; i++) s.offsets[i] = w.offsets[start + i] - b0;
        s.data.assign(w.data.begin() + b0, w.data.begin() + b1);
        ch.bytes += (cnt + 1) * 4 + (b1 - b0);
      } else {
        const int wd = type_width(g->types[input][c]);
        s.data.assign(w.data.begin() + start * wd, w.data.begin() + (start + cnt) * wd);
        ch.bytes += cnt * wd;
      }
      if (!w.present.empty()) {
        bool any_null = false;
        for (int64_t i = 0; i < cnt && !any_null; i++) any_null = w.present[start + i] == 0;
        if (any_null) {
          s.present.assign(w.present.begin() + start, w.present.begin() + start + cnt);
          ch.bytes += cnt;
        }
      }
    }
    const int idx = g->part_offset + q;
    PartInfo &pi = g->main_parts[idx];
    if (input == g->ninputs - 1) {
      pi.parent_mem = parent_mem;
      pi.mem_size += ch.bytes;
    }
    pi.rows[input] += cnt;
    g->stats.bytes_spilled += ch.bytes;
    if (g->host_bytes + ch.bytes > g->host_limit && !g->spill_dir.empty()) {
      CRGPU_TRY(chunk_to_disk(g, ch));
    } else {
      g->host_bytes += ch.bytes;
    }
    pi.chunks[input].push_back(std::move(ch));
    start += cnt;
  }
  return CRGPU_OK;
}

// host chunk(s) -> device columns of one input
static crgpu_status upload_chunks(crgpu_grace *g, int input, std::vector<SpillChunk> &chunks, LoadedInput &L) {
  const int nc = static_cast<int>(g->types[input].size());
  int64_t n = 0;
  for (SpillChunk &c : chunks) {
    CRGPU_TRY(chunk_fetch(g, c));
    n += c.n;
  }
  L.n = n;
  if (static_cast<int>(L.data.size()) != nc) {  // (DevBuf is neither copyable nor movable: whole vectors are replaced)
    L.data = std::vector<DevBuf>(nc);
    L.valid = std::vector<DevBuf>(nc);
    L.offsets = std::vector<DevBuf>(nc);
  }
  L.cols.resize(nc);
  hipStream_t st = stream();
  for (int c = 0; c < nc; c++) {
    const int t = g->types[input][c];
    crgpu_dcol &d = L.cols[c];
    d.type = t;
    d._pad = 0;
    d.valid = nullptr;
    d.offsets = nullptr;
    size_t total = 0;
    bool any_null = false;
    for (const SpillChunk &ch : chunks) {
      total += ch.cols[c].data.size();
      any_null = any_null || !ch.cols[c].present.empty();
    }
    CRGPU_TRY(L.data[c].ensure(total + 16));
    d.data = L.data[c].p;
    size_t at = 0;
    for (const SpillChunk &ch : chunks) {
      const size_t sz = ch.cols[c].data.size();
      if (sz) CRGPU_HIP(hipMemcpyAsync(static_cast<uint8_t *>(L.data[c].p) + at, ch.cols[c].data.data(), sz, hipMemcpyHostToDevice, st));
      at += sz;
    }
    if (t == CRGPU_BYTES) {
      if (total > 0xFFFFFFF0ull) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_grace: a partition's BYTES column exceeds 4 GiB");
      std::vector<uint32_t> offs(static_cast<size_t>(n) + 1);
      size_t row = 0;
      uint32_t base = 0;
      offs[0] = 0;
      for (const SpillChunk &ch : chunks) {
        for (int64_t i = 1; i <= ch.n; i++) offs[row + i] = base + ch.cols[c].offsets[i];
        row += ch.n;
        base += ch.cols[c].offsets[ch.n];
      }
      CRGPU_TRY(L.offsets[c].ensure(offs.size() * 4));
      CRGPU_HIP(hipMemcpyAsync(L.offsets[c].p, offs.data(), offs.size() * 4, hipMemcpyHostToDevice, st));
      CRGPU_HIP(hipStreamSynchronize(st));  // offs is a temporary
      d.offsets = L.offsets[c].as<uint32_t>();
    }
    if (any_null) {
      std::vector<uint8_t> bm(static_cast<size_t>((n + 7) / 8) + 1, 0);
      int64_t row = 0;
      for (const SpillChunk &ch : chunks) {
        const std::vector<uint8_t> &pr = ch.cols[c].present;
        for (int64_t i = 0; i < ch.n; i++)
          if (pr.empty() || pr[i]) bm[(row + i) >> 3] |= static_cast<uint8_t>(1u << ((row + i) & 7));
        row += ch.n;
      }
      CRGPU_TRY(L.valid[c].ensure(bm.size()));
      CRGPU_HIP(hipMemcpyAsync(L.valid[c].p, bm.data(), bm.size(), hipMemcpyHostToDevice, st));
      CRGPU_HIP(hipStreamSynchronize(st));
      d.valid = L.valid[c].as<uint8_t>();
    }
  }
  CRGPU_HIP(hipStreamSynchronize(st));  // the chunks' host vectors are released by the caller
  return CRGPU_OK;
}

static crgpu_status grace_common(crgpu_grace *g, const crgpu_grace_opts *o) {
  g->mem_limit = o != nullptr && o->mem_limit > 0 ? o->mem_limit : int64_t(64) << 20;  // sql.distsql.temp_storage.workmem default
  if (g->mem_limit < kMinimalMaxPartitionSize) g->mem_limit = kMinimalMaxPartitionSize;
  if (o != nullptr && o->num_buckets > 0) g->num_buckets = o->num_buckets;
  if (g->num_buckets < 2 || g->num_buckets > 1024) return fail(CRGPU_ERR_INVALID, "crgpu_grace: between 2 and 1024 buckets");
  if (o != nullptr && o->max_repartitions > 0) g->max_repartitions = o->max_repartitions;
  if (o != nullptr && o->host_limit > 0) g->host_limit = o->host_limit;
  if (o != nullptr && o->spill_dir != nullptr) g->spill_dir = o->spill_dir;
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_grace_join_create(const crgpu_join_spec *spec, const crgpu_grace_opts *opts, crgpu_grace **out) {
  CRGPU_TRY(ensure_init());
  if (spec == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_grace_join_create: null argument");
  std::unique_ptr<crgpu_grace> g(new crgpu_grace());
  CRGPU_TRY(g->open());
  CRGPU_TRY(grace_common(g.get(), opts));
  CRGPU_TRY(crgpu_dev_join_create(spec, &g->join));
  g->kind = 0;
  g->ninputs = 2;
  g->join_type = spec->join_type;
  g->types[0].assign(spec->left_types, spec->left_types + spec->n_left_cols);
  g->types[1].assign(spec->right_types, spec->right_types + spec->n_right_cols);
  g->hash_cols[0].assign(spec->left_eq, spec->left_eq + spec->n_eq);
  g->hash_cols[1].assign(spec->right_eq, spec->right_eq + spec->n_eq);
  *out = g.release();
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_grace_agg_create(const crgpu_agg_spec *spec, const crgpu_grace_opts *opts, crgpu_grace **out) {
  CRGPU_TRY(ensure_init());
  if (spec == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_grace_agg_create: null argument");
  // (external_hash_aggregator.go:88-118 restores the ordering with a sort behind the partitioned aggregation)
  if (spec->n_ordered != 0) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_grace_agg_create: ordered grouping columns: hash partitions do not keep the input order");
  std::unique_ptr<crgpu_grace> g(new crgpu_grace());
  CRGPU_TRY(g->open());
  CRGPU_TRY(grace_common(g.get(), opts));
  CRGPU_TRY(crgpu_dev_agg_create(spec, &g->agg));
  g->kind = 1;
  g->ninputs = 1;
  g->types[0].assign(spec->types, spec->types + spec->n_cols);
  g->hash_cols[0].assign(spec->group_cols, spec->group_cols + spec->n_group);
  *out = g.release();
  return CRGPU_OK;
}

extern "C" void crgpu_dev_grace_destroy(crgpu_grace *g) {
  if (g == nullptr) return;
  if (g->st != nullptr) (void)hipStreamSynchronize(g->st);
  delete g;
}

extern "C" crgpu_status crgpu_dev_grace_add(crgpu_grace *g, int32_t input, const crgpu_dcol *cols, int64_t n) {
  if (g == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_grace_add: null handle");
  CRGPU_ENTER(g);
  CRGPU_CHECK_CANCEL(g);
  if (!g->partitioning) return fail(CRGPU_ERR_STATE, "crgpu_dev_grace_add: the partitioning phase is over (crgpu_dev_grace_next was called)");
  if (input < 0 || input >= g->ninputs) return fail(CRGPU_ERR_INVALID, "crgpu_dev_grace_add: input ordinal out of range");
  if (n < 0) return fail(CRGPU_ERR_INVALID, "crgpu_dev_grace_add: negative row count");
  if (n > 0 && cols == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_grace_add: null columns");
  return partition_and_spill(g, input, cols, n, INT64_MAX);
}

// hbpRecursivePartitioning for one parent partition (:482-560)
static crgpu_status repartition(crgpu_grace *g, int parent_idx) {
  if (++g->stats.repartitions > g->max_repartitions)
    return fail(CRGPU_ERR_OOM, "crgpu_grace: the partitions do not shrink (max_repartitions reached)");
  g->seed++;  // "in order to use a different hash function when repartitioning" (:489-491)
  PartInfo parent = std::move(g->main_parts[parent_idx]);
  g->main_parts.erase(parent_idx);
  for (int i = 0; i < g->ninputs; i++) {
    for (SpillChunk &ch : parent.chunks[i]) {  // chunk by chunk: a parent never has to fit HBM as a whole
      std::vector<SpillChunk> one;
      one.push_back(std::move(ch));
      LoadedInput L;
      CRGPU_TRY(upload_chunks(g, i, one, L));
      CRGPU_TRY(partition_and_spill(g, i, L.cols.data(), L.n, parent.mem_size));
    }
  }
  for (int q = 0; q < g->num_buckets; q++) {
    const int idx = g->part_offset + q;
    auto it = g->main_parts.find(idx);
    if (it == g->main_parts.end()) continue;
    const int64_t before = it->second.parent_mem, after = it->second.mem_size;
    if (before > 0 && before != INT64_MAX) {
      const double decrease = 1.0 - static_cast<double>(after) / static_cast<double>(before);
      if (decrease < kSizeDecreaseThreshold) {  // rows not distinct on the hash columns: repartitioning cannot help
        g->fallback_parts[idx] = std::move(it->second);
        g->fallback_order.push_back(idx);
        g->main_parts.erase(it);
      }
    }
  }
  g->part_offset += g->num_buckets;
  return CRGPU_OK;
}

static crgpu_status run_partition(crgpu_grace *g, PartInfo &pi, crgpu_dev_result *out) {
  for (int i = 0; i < g->ninputs; i++) CRGPU_TRY(upload_chunks(g, i, pi.chunks[i], g->loaded[i]));
  for (int i = 0; i < g->ninputs; i++) pi.chunks[i].clear();
  g->stats.partitions_processed++;
  if (g->kind == 1) {
    CRGPU_TRY(crgpu_dev_agg_reset(g->agg));
    CRGPU_TRY(crgpu_dev_agg_consume(g->agg, g->loaded[0].cols.data(), g->loaded[0].n));
    return crgpu_dev_agg_finish(g->agg, out);
  }
  CRGPU_TRY(crgpu_dev_join_reset(g->join));
  CRGPU_TRY(crgpu_dev_join_build(g->join, g->loaded[1].cols.data(), g->loaded[1].n));
  CRGPU_TRY(crgpu_dev_join_probe(g->join, g->loaded[0].cols.data(), g->loaded[0].n, out));
  g->emit_right_pending = g->join_type == CRGPU_RIGHT_OUTER || g->join_type == CRGPU_FULL_OUTER ||
                          g->join_type == CRGPU_RIGHT_SEMI || g->join_type == CRGPU_RIGHT_ANTI;
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_grace_next(crgpu_grace *g, crgpu_dev_result *out, int32_t *done) {
  if (g == nullptr || out == nullptr || done == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_grace_next: null argument");
  CRGPU_ENTER(g);
  CRGPU_CHECK_CANCEL(g);
  *done = 0;
  out->n = 0;
  if (g->partitioning) {  // "all inputs have been partitioned and spilled" (:447)
    g->partitioning = false;
    g->part_offset += g->num_buckets;
  }
  while (true) {
    CRGPU_CHECK_CANCEL(g);
    if (g->finished) {
      *done = 1;
      out->n = 0;
      out->ncols = 0;
      out->cols = nullptr;
      return CRGPU_OK;
    }
    if (g->emit_right_pending) {  // hjEmittingRight of the partition in HBM
      g->emit_right_pending = false;
      CRGPU_TRY(crgpu_dev_join_emit_right(g->join, out));
      if (out->n > 0) return CRGPU_OK;
      continue;
    }
    // hbpProcessNewPartitionUsingMain (:562-611)
    int pick = -1;
    for (auto &kv : g->main_parts)
      if (kv.second.mem_size <= g->mem_limit) {
        pick = kv.first;
        break;
      }
    if (pick >= 0) {
      PartInfo pi = std::move(g->main_parts[pick]);
      g->main_parts.erase(pick);
      CRGPU_TRY(run_partition(g, pi, out));
      if (out->n > 0) return CRGPU_OK;
      continue;
    }
    if (!g->main_parts.empty()) {
      CRGPU_TRY(repartition(g, g->main_parts.begin()->first));
      continue;
    }
    if (!g->fallback_order.empty()) {  // hbpProcessNewPartitionUsingFallback (:628-643): see the file header
      const int idx = g->fallback_order.front();
      g->fallback_order.erase(g->fallback_order.begin());
      PartInfo pi = std::move(g->fallback_parts[idx]);
      g->fallback_parts.erase(idx);
      g->stats.fallback_partitions++;
      CRGPU_TRY(run_partition(g, pi, out));
      if (out->n > 0) return CRGPU_OK;
      continue;
    }
    g->finished = true;
  }
}

extern "C" crgpu_status crgpu_dev_grace_stats(crgpu_grace *g, crgpu_grace_stats *out) {
  if (g == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_grace_stats: null argument");
  *out = g->stats;
  return CRGPU_OK;
}
