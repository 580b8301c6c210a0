/* TEST INFRASTRUCTURE ONLY — CPU restatement (plain C) of the reference's integer algorithms on the
 * sparse-convolution hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this; the product (minkowskiengine_amd/) never does.
 *
 * Every function cites the reference code it follows (paths under the MinkowskiEngine tree).
 * Pinned against (a) the reference's own golden vectors (tests/cpp/kernel_region_cpu_test.py,
 * tests/cpp/coordinate_map_cpu_test.py, tests/python/coordinate_manager.py) in
 * tests/test_oracle_golden.py and (b) outputs of the reference itself (oracle/_ref/_C.so, built
 * unmodified by oracle/build_ref.py) in tests/test_oracle_vs_reference.py and tests/golden/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_DIM 7
#define ORC_HYPER_CUBE 0
#define ORC_HYPER_CROSS 1

typedef struct orc_region {
  int32_t ncol;        /* D + 1 */
  int32_t region_type; /* src/types.hpp RegionType */
  int32_t kernel_size[ORC_MAX_DIM];
  int32_t dilation[ORC_MAX_DIM];
  int32_t tensor_stride[ORC_MAX_DIM];
} orc_region;

/* ---- a small open-addressing hash map coordinate -> row (stands in for robin_hood,
 *      src/coordinate_map_cpu.hpp:75-86; iteration order is never used here) ------------------- */
typedef struct orc_map {
  const int32_t *coords; /* [n, ncol], rows referenced by value */
  int32_t ncol;
  int64_t cap; /* power of two */
  int32_t *slot_row; /* -1 = empty */
} orc_map;

static uint64_t orc_hash(const int32_t *c, int ncol) {
  uint64_t h = 1469598103934665603ull; /* FNV-1a over the raw bytes */
  const unsigned char *p = (const unsigned char *)c;
  for (int i = 0; i < ncol * 4; ++i) {
    h ^= p[i];
    h *= 1099511628211ull;
  }
  h ^= h >> 29;
  return h;
}

static int orc_map_init(orc_map *m, const int32_t *coords, int64_t n, int ncol) {
  int64_t cap = 16;
  while (cap < 2 * n) cap <<= 1;
  m->coords = coords;
  m->ncol = ncol;
  m->cap = cap;
  m->slot_row = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
  if (!m->slot_row) return 1;
  memset(m->slot_row, 0xff, (size_t)cap * sizeof(int32_t));
  return 0;
}
static void orc_map_free(orc_map *m) { free(m->slot_row); }

/* returns the row holding `key` or -1; if insert_row >= 0 and the key is absent, stores it */
static int32_t orc_map_find_or_insert(orc_map *m, const int32_t *key, int32_t insert_row) {
  uint64_t pos = orc_hash(key, m->ncol) & (uint64_t)(m->cap - 1);
  for (;;) {
    const int32_t r = m->slot_row[pos];
    if (r < 0) {
      if (insert_row >= 0) m->slot_row[pos] = insert_row;
      return -1;
    }
    if (memcmp(m->coords + (int64_t)r * m->ncol, key, (size_t)m->ncol * 4) == 0) return r;
    pos = (pos + 1) & (uint64_t)(m->cap - 1);
  }
}

/* CoordinateMapCPU::insert_and_map<true> (src/coordinate_map_cpu.hpp:353-380): sequential inserts,
 * the first occurrence of a coordinate wins, value = running count of unique rows.
 *   unique_map [n] (first n_unique valid), inverse_map [n];  returns n_unique (or -1). */
int64_t orc_insert_and_map(const int32_t *coords, int64_t n, int32_t ncol, int64_t *unique_map,
                           int64_t *inverse_map) {
  /* the map stores rows of `coords`; value (unique index) is kept in a side array */
  orc_map m;
  if (orc_map_init(&m, coords, n, ncol)) return -1;
  int32_t *value_of_row = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  int64_t value = 0;
  for (int64_t row = 0; row < n; ++row) {
    const int32_t *key = coords + row * ncol;
    const int32_t found = orc_map_find_or_insert(&m, key, (int32_t)row);
    if (found < 0) { /* result.second == true */
      unique_map[value] = row;
      inverse_map[row] = value;
      value_of_row[row] = (int32_t)value;
      value += 1;
    } else {
      inverse_map[row] = value_of_row[found];
    }
  }
  free(value_of_row);
  orc_map_free(&m);
  return value;
}

/* detail::stride_coordinate (src/coordinate_map.hpp:58-66): dst = floor((float)src / ts) * ts,
 * batch index copied; ts is the OUTPUT tensor stride (coordinate_map_cpu.hpp:429-431). */
void orc_stride_coordinates(const int32_t *coords, int64_t n, int32_t ncol, const int32_t *out_ts,
                            int32_t *out) {
  for (int64_t i = 0; i < n; ++i) {
    out[i * ncol] = coords[i * ncol];
    for (int d = 0; d < ncol - 1; ++d)
      out[i * ncol + d + 1] =
          (int32_t)(floorf((float)coords[i * ncol + d + 1] / (float)out_ts[d]) * (float)out_ts[d]);
  }
}

/* cpu_kernel_region::set_volume (src/kernel_region.hpp:250-270) */
int64_t orc_region_volume(const orc_region *rg) {
  int64_t v = 1;
  if (rg->region_type == ORC_HYPER_CUBE) {
    for (int i = 0; i < rg->ncol - 1; ++i) v *= rg->kernel_size[i];
  } else {
    for (int i = 0; i < rg->ncol - 1; ++i) v += rg->kernel_size[i] - 1;
  }
  return v;
}

/* cpu_kernel_region::coordinate_at (src/kernel_region.hpp:198-247) */
void orc_coordinate_at(const orc_region *rg, int32_t kernel_index, const int32_t *src, int32_t *dst) {
  dst[0] = src[0];
  if (rg->region_type == ORC_HYPER_CUBE) {
    for (int i = 0; i < rg->ncol - 1; ++i) {
      const int32_t ks = rg->kernel_size[i];
      const int32_t cur = kernel_index % ks;
      if (ks % 2 == 0)
        dst[i + 1] = src[i + 1] + rg->dilation[i] * rg->tensor_stride[i] * cur;
      else
        dst[i + 1] = src[i + 1] + (cur - ks / 2) * rg->dilation[i] * rg->tensor_stride[i];
      kernel_index /= ks;
    }
  } else { /* HYPER_CROSS */
    for (int i = 1; i < rg->ncol; ++i) dst[i] = src[i];
    if (kernel_index == 0) return;
    int32_t ind = kernel_index - 1;
    int axis = 0;
    while (axis < rg->ncol - 1) {
      if (ind < rg->kernel_size[axis] - 1) break;
      ind -= rg->kernel_size[axis] - 1;
      axis += 1;
    }
    const int32_t r = (rg->kernel_size[axis] - 1) / 2;
    const int32_t off = (ind < r) ? (ind + 1) : (ind - 2 * r);
    dst[axis + 1] += off * rg->dilation[axis] * rg->tensor_stride[axis];
  }
}

/* the region iterator of the reference tests (tests/cpp/kernel_region_cpu_test.py region_iterator_test):
 * out [n * volume, ncol] = all neighbour coordinates of every input coordinate, offset-major per point */
void orc_region_coordinates(const orc_region *rg, const int32_t *coords, int64_t n, int32_t *out) {
  const int64_t vol = orc_region_volume(rg);
  for (int64_t i = 0; i < n; ++i)
    for (int64_t k = 0; k < vol; ++k)
      orc_coordinate_at(rg, (int32_t)k, coords + i * rg->ncol, out + (i * vol + k) * rg->ncol);
}

/* CoordinateMapCPU::kernel_map (src/coordinate_map_cpu.hpp:569-670): for every OUTPUT row u and
 * every offset k, look coordinate_at(k, out[u]) up in the INPUT map; hit -> pair (in row, out row)
 * in list k.  The reference fills the lists from OpenMP threads in nondeterministic order; here the
 * order is by output row, and parity is defined on the per-offset pair SETS.
 *   nbr [volume, n_out] = in row or -1;  counts [volume].  Returns total pairs (or -1). */
int64_t orc_kernel_map(const int32_t *in_coords, int64_t n_in, const int32_t *out_coords, int64_t n_out,
                       const orc_region *rg, int32_t *nbr, int64_t *counts) {
  orc_map m;
  const int ncol = rg->ncol;
  if (orc_map_init(&m, in_coords, n_in, ncol)) return -1;
  for (int64_t r = 0; r < n_in; ++r) orc_map_find_or_insert(&m, in_coords + r * ncol, (int32_t)r);
  const int64_t vol = orc_region_volume(rg);
  int64_t total = 0;
  int32_t key[ORC_MAX_DIM + 1];
  for (int64_t k = 0; k < vol; ++k) counts[k] = 0;
  for (int64_t u = 0; u < n_out; ++u) {
    for (int64_t k = 0; k < vol; ++k) {
      orc_coordinate_at(rg, (int32_t)k, out_coords + u * ncol, key);
      const int32_t r = orc_map_find_or_insert(&m, key, -1);
      nbr[k * n_out + u] = r;
      if (r >= 0) {
        counts[k] += 1;
        total += 1;
      }
    }
  }
  orc_map_free(&m);
  return total;
}

/* rows[q] = row of query q in the map or -1 (CoordinateMapCPU::find, coordinate_map_cpu.hpp:388-412) */
int orc_find(const int32_t *map_coords, int64_t n, int32_t ncol, const int32_t *queries, int64_t nq,
             int32_t *rows) {
  orc_map m;
  if (orc_map_init(&m, map_coords, n, ncol)) return 1;
  for (int64_t r = 0; r < n; ++r) orc_map_find_or_insert(&m, map_coords + r * ncol, (int32_t)r);
  for (int64_t q = 0; q < nq; ++q) rows[q] = orc_map_find_or_insert(&m, queries + q * ncol, -1);
  orc_map_free(&m);
  return 0;
}
