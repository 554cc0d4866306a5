/* CPU ORACLE (plain C) for the non-encoding rows of the PermutoSDF hot path.  TEST INFRASTRUCTURE ONLY:
 * imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product path.
 *
 * Each function restates, item by item and in the reference's order of floating-point operations (fp32, one
 * rounding per operation, no FMA contraction: build with -ffp-contract=off), one CUDA kernel of
 * /root/reference/kernels/permuto_sdf/*.cuh; the file:line it follows is cited on every function.
 * PINNED: tests/test_oracle_vs_ref.py checks every function here bit-for-bit against the reference's own
 * kernel headers compiled for the CPU (oracle/_ref/libpsdf_ref.so, built by `make ref`), and
 * tests/golden/*.npz holds vectors generated from that reference build (tests/golden/make_golden.py).
 *
 * Semantics kept from the reference: slot reservation through a running counter (the reference uses atomicAdd;
 * executed sequentially here, so ranges are ray ordered), holes + z=-1 sentinels when a ray emits fewer samples
 * than it reserved, rays with <= 2 samples dropped, silent skip of rays whose reservation exceeds the pool.
 * One deliberate difference: float->uint32 conversion of voxel coordinates saturates (negative -> 0), which is
 * what the GPU does and what the reference relies on; a host build of the reference wraps.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
static v3 mk3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static v3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
static void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static v3 add3(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 sub3(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 scale3(float s, v3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
static float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 ray_at(v3 o, float t, v3 d) { return add3(o, scale3(t, d)); }
static float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
static int clampi(int f, int a, int b) { return f < a ? a : (f > b ? b : f); }

/* ------------------------------------------------------------------ pcg32 (pcg32.h:45-206) */
typedef struct { uint64_t state, inc; } pcg;
#define PCG_MULT 0x5851f42d4c957f2dULL
static uint32_t pcg_next_uint(pcg* r) {
  uint64_t old = r->state;
  r->state = old * PCG_MULT + r->inc;
  uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  uint32_t rot = (uint32_t)(old >> 59u);
  return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
static float pcg_next_float(pcg* r) {
  union { uint32_t u; float f; } c;
  c.u = (pcg_next_uint(r) >> 9) | 0x3f800000u;
  return c.f - 1.0f;
}
static void pcg_advance(pcg* r, int64_t delta_) {
  uint64_t cur_mult = PCG_MULT, cur_plus = r->inc, acc_mult = 1u, acc_plus = 0u, delta = (uint64_t)delta_;
  while (delta > 0) {
    if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
    cur_plus = (cur_mult + 1) * cur_plus;
    cur_mult *= cur_mult;
    delta /= 2;
  }
  r->state = acc_mult * r->state + acc_plus;
}
void orc_pcg32(uint64_t* state, uint64_t* inc, int64_t advance, int n, uint32_t* out_u, float* out_f) {
  pcg r = {*state, *inc};
  if (advance) pcg_advance(&r, advance);
  for (int i = 0; i < n; i++) {
    if (out_u) out_u[i] = pcg_next_uint(&r);
    if (out_f) out_f[i] = pcg_next_float(&r);
  }
  *state = r.state;
  *inc = r.inc;
}

/* ------------------------------------------------------------------ Morton grid (OccupancyGridGPU.cuh:37-193) */
static uint32_t spread10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
uint32_t orc_morton3D(uint32_t x, uint32_t y, uint32_t z) { return spread10(x) | (spread10(y) << 1) | (spread10(z) << 2); }
uint32_t orc_morton3D_invert(uint32_t x) {
  x = x & 0x49249249;
  x = (x | (x >> 2)) & 0xc30c30c3;
  x = (x | (x >> 4)) & 0x0f00f00f;
  x = (x | (x >> 8)) & 0xff0000ff;
  x = (x | (x >> 16)) & 0x0000ffff;
  return x;
}
static uint32_t f2u_sat(float f) {
  if (!(f > 0.f)) return 0u;
  if (f >= 4294967296.f) return 0xFFFFFFFFu;
  return (uint32_t)f;
}
typedef struct { int n; float extent; v3 tr; } grid_t;
/* lin_idx_to_3D (:112-155), centre of voxel */
static v3 voxel_centre(uint32_t idx, grid_t g) {
  float x = (float)orc_morton3D_invert(idx), y = (float)orc_morton3D_invert(idx >> 1), z = (float)orc_morton3D_invert(idx >> 2);
  x = x / g.n; y = y / g.n; z = z / g.n;
  x = (float)(x - 0.5); y = (float)(y - 0.5); z = (float)(z - 0.5);
  float voxel = (float)(1.0 / g.n);
  float half = voxel / 2;
  x += half; y += half; z += half;
  x = x * g.extent; y = y * g.extent; z = z * g.extent;
  return mk3(x + g.tr.x, y + g.tr.y, z + g.tr.z);
}
/* pos_to_lin_idx (:158-193) with get_center_of_voxel=false */
static int voxel_of(v3 p, grid_t g) {
  float x = p.x - g.tr.x, y = p.y - g.tr.y, z = p.z - g.tr.z;
  x = x / g.extent; y = y / g.extent; z = z / g.extent;
  x = (float)(x + 0.5); y = (float)(y + 0.5); z = (float)(z + 0.5);
  x = x * g.n; y = y * g.n; z = z * g.n;
  return (int)orc_morton3D(f2u_sat(x), f2u_sat(y), f2u_sat(z));
}
static int voxel_oob(int v, grid_t g) { return v >= g.n * g.n * g.n || v < 0; }
static int sgnf(float x) { int t = x < 0 ? -1 : 0; return x > 0 ? 1 : t; }
/* distance_to_next_voxel (:95-109) */
static float next_voxel_dist(v3 pos, v3 dir, v3 idir, int n) {
  pos = scale3((float)n, pos);
  float tx = (floorf(pos.x + 0.5f + 0.5f * sgnf(dir.x)) - pos.x) * idir.x;
  float ty = (floorf(pos.y + 0.5f + 0.5f * sgnf(dir.y)) - pos.y) * idir.y;
  float tz = (floorf(pos.z + 0.5f + 0.5f * sgnf(dir.z)) - pos.z) * idir.z;
  float t = fminf(fminf(fabsf(tx), fabsf(ty)), fabsf(tz));
  return fmaxf(t / n, 0.0f);
}
static v3 inv_dir(v3 d) {
  v3 r;
  r.x = fabs(d.x) < 1e-16 ? 0.f : (float)(1.0 / d.x);
  r.y = fabs(d.y) < 1e-16 ? 0.f : (float)(1.0 / d.y);
  r.z = fabs(d.z) < 1e-16 ? 0.f : (float)(1.0 / d.z);
  return r;
}
static grid_t mk_grid(int n, float extent, const float* tr) { grid_t g = {n, extent, {tr[0], tr[1], tr[2]}}; return g; }

/* compute_grid_points_gpu (:196) / compute_random_sample_of_grid_points_gpu (:248) */
void orc_grid_points(int count, int n, float extent, const float* tr, const int* indices, uint64_t st, uint64_t inc,
                     int randomize, float* out) {
  grid_t g = mk_grid(n, extent, tr);
  for (int i = 0; i < count; i++) {
    v3 p = voxel_centre(indices ? (uint32_t)indices[i] : (uint32_t)i, g);
    if (randomize) {
      float voxel = extent / n;
      float half = (float)(voxel / 2.0);
      pcg r = {st, inc};
      pcg_advance(&r, i * 3);
      float rnd = pcg_next_float(&r);
      p.x += voxel * rnd - half;
      rnd = pcg_next_float(&r);
      p.y += voxel * rnd - half;
      rnd = pcg_next_float(&r);
      p.z += voxel * rnd - half;
    }
    st3(out + 3 * (int64_t)i, p);
  }
}
/* update_with_density_gpu (:303) / _random_sample_gpu (:340) */
void orc_update_with_density(int count, const int* indices, const float* density, float decay, float thresh,
                             float* values, uint8_t* occ) {
  for (int i = 0; i < count; i++) {
    int v = indices ? indices[i] : i;
    float old = values[v] * decay;
    float upd = fmaxf(density[i], old);
    values[v] = upd;
    occ[v] = upd > thresh;
  }
}
/* logistic_density_distribution (:381) */
static float logistic_density(float x, float s) { return s * expf(-s * x) / (powf((1 + expf(-s * x)), 2)); }
/* update_with_sdf_gpu (:387, range 1.3) / update_with_sdf_random_sample_gpu (:447, range 1.0, inv_s from a tensor) */
void orc_update_with_sdf(int count, const int* indices, const float* sdf, float extent, int n, float inv_s,
                         const float* inv_s_tensor, float thresh, float* values, uint8_t* occ) {
  for (int i = 0; i < count; i++) {
    int v = indices ? indices[i] : i;
    float voxel = extent / n;
    float half = (float)(voxel / 2.0);
    float half_diag = sqrtf(3.0) * half;
    float s_new = sdf[i];
    values[v] = s_new;
    float range = (float)((indices ? 1.0 : 1.3) * half_diag);
    float lo = fabs(s_new) - range;
    float capped = clampf(lo, 0.0, 1e10);
    float w = logistic_density(capped, indices ? inv_s_tensor[0] : inv_s);
    occ[v] = w > thresh;
  }
}
/* check_occupancy_gpu (:901) */
void orc_check_occupancy(int count, int n, float extent, const float* tr, const uint8_t* occ, const float* pts, uint8_t* out) {
  grid_t g = mk_grid(n, extent, tr);
  for (int i = 0; i < count; i++) {
    int v = voxel_of(ld3(pts + 3 * (int64_t)i), g);
    out[i] = voxel_oob(v, g) ? 0 : occ[v];
  }
}

#define MAX_STEPS 4096
/* compute_samples_in_occupied_regions_gpu (:510-703); use_grid=0: compute_samples_fg_gpu (RaySamplerGPU.cuh:162-335) */
void orc_march_samples(int use_grid, int R, int n, float extent, const float* tr, const uint8_t* occ, const float* origins,
                       const float* dirs, const float* t_entry, const float* t_exit_p, float min_dist, int max_per_ray,
                       int max_nr_samples, uint64_t st, uint64_t inc, int jitter, float* s_pos, float* s_dirs, float* s_z,
                       float* s_dt, float* fixed_dt, int* start_end, int* cur_nr_samples) {
  grid_t g;
  if (use_grid) g = mk_grid(n, extent, tr); else { float z[3] = {0, 0, 0}; g = mk_grid(1, 1.f, z); }
  const float eps = 1e-6;
  for (int ray = 0; ray < R; ray++) {
    v3 org = ld3(origins + 3 * (int64_t)ray), dir = ld3(dirs + 3 * (int64_t)ray), idir = inv_dir(dir);
    float t_start = t_entry[ray], t_exit = t_exit_p[ray];
    float t = t_start, occupied = 0.0;
    int steps = 0;
    if (use_grid) {
      while (t < t_exit && steps < MAX_STEPS) {
        v3 pos = ray_at(org, t, dir);
        int v = voxel_of(pos, g);
        if (voxel_oob(v, g)) break;
        float d = next_voxel_dist(pos, dir, idir, g.n);
        t += d;
        t += eps;
        if (occ[v]) {
          occupied += d;
          if ((t - eps) > t_exit) occupied -= (t - eps) - t_exit;
        }
        steps += 1;
      }
    } else {
      occupied = t_exit - t_start;
    }
    int to_create = (int)(occupied / min_dist);
    to_create = clampi(to_create, 0, max_per_ray);
    float spacing = occupied / to_create;
    int go = use_grid ? (to_create > 1) : (to_create > 1 && occupied > eps);
    if (!go) {
      fixed_dt[ray] = 0;
      start_end[2 * ray] = 0;
      start_end[2 * ray + 1] = 0;
      continue;
    }
    int base = cur_nr_samples[0];
    cur_nr_samples[0] += to_create;
    start_end[2 * ray] = base;
    start_end[2 * ray + 1] = base + to_create;
    fixed_dt[ray] = spacing;
    if (base + to_create > max_nr_samples) continue;
    t = t_start;
    steps = 0;
    pcg rng = {st, inc};
    if (jitter) {
      pcg_advance(&rng, ray);
      t = t + spacing * pcg_next_float(&rng);
    }
    int created = 0;
    while (t < t_exit && steps < MAX_STEPS) {
      t = clampf(t, t_start, t_exit);
      v3 pos = ray_at(org, t, dir);
      int occupied_here = 1;
      if (use_grid) {
        int v = voxel_of(pos, g);
        if (voxel_oob(v, g)) break;
        occupied_here = occ[v];
      }
      if (occupied_here && created < to_create) {
        int64_t o = base + created;
        st3(s_pos + 3 * o, pos);
        st3(s_dirs + 3 * o, dir);
        s_z[o] = t;
        s_dt[o] = spacing;
        t += spacing;
        created += 1;
      } else if (use_grid) {
        float delta = next_voxel_dist(pos, dir, idir, g.n);
        if (jitter) delta = delta + spacing * pcg_next_float(&rng);
        t += delta;
        t += eps;
      }
      steps += 1;
    }
    if (created > 0) { /* guard of the reference's out-of-range access when nothing was created (App. B5) */
      float remaining = t_exit - s_z[base + created - 1];
      s_dt[base + created - 1] = clampf(remaining, 0.0, spacing);
    }
    for (int i = created; i < to_create; i++) {
      int64_t o = base + i;
      st3(s_pos + 3 * o, mk3(0, 0, 0));
      st3(s_dirs + 3 * o, mk3(0, 0, 0));
      s_z[o] = -1;
      s_dt[o] = 0;
    }
    start_end[2 * ray + 1] = base + created;
    if (created <= 2) {
      fixed_dt[ray] = 0;
      start_end[2 * ray] = 0;
      start_end[2 * ray + 1] = 0;
    }
  }
}
/* compute_first_sample_start_of_occupied_regions_gpu (:707-814) */
void orc_first_hit_samples(int R, int n, float extent, const float* tr, const uint8_t* occ, const float* origins,
                           const float* dirs, const float* t_entry, const float* t_exit_p, int max_nr_samples,
                           float* s_pos, float* s_dirs, float* s_z, float* s_dt, float* fixed_dt, int* start_end,
                           int* cur_nr_samples) {
  grid_t g = mk_grid(n, extent, tr);
  const float eps = 1e-6;
  for (int ray = 0; ray < R; ray++) {
    v3 org = ld3(origins + 3 * (int64_t)ray), dir = ld3(dirs + 3 * (int64_t)ray), idir = inv_dir(dir);
    float t = t_entry[ray], t_exit = t_exit_p[ray];
    int hit = 0;
    while (t < t_exit) {
      v3 pos = ray_at(org, t, dir);
      int v = voxel_of(pos, g);
      if (voxel_oob(v, g)) break;
      float d = next_voxel_dist(pos, dir, idir, g.n);
      t += d;
      t += eps;
      if (occ[v]) {
        int base = cur_nr_samples[0];
        cur_nr_samples[0] += 1;
        start_end[2 * ray] = base;
        start_end[2 * ray + 1] = base + 1;
        fixed_dt[ray] = 0;
        hit = 1;
        if (base + 1 > max_nr_samples) break;
        st3(s_pos + 3 * (int64_t)base, pos);
        st3(s_dirs + 3 * (int64_t)base, dir);
        s_z[base] = t;
        s_dt[base] = 0;
        break;
      }
    }
    if (!hit) {
      fixed_dt[ray] = 0;
      start_end[2 * ray] = 0;
      start_end[2 * ray + 1] = 0;
    }
  }
}
/* advance_sample_to_next_occupied_voxel_gpu (:817-895) */
void orc_advance_samples(int count, int n, float extent, const float* tr, const uint8_t* occ, const float* dirs,
                         const float* pos_in, float* pos_out, uint8_t* within) {
  grid_t g = mk_grid(n, extent, tr);
  const float eps = 1e-6;
  for (int i = 0; i < count; i++) {
    v3 org = ld3(pos_in + 3 * (int64_t)i), dir = ld3(dirs + 3 * (int64_t)i), idir = inv_dir(dir);
    float t = 0;
    int steps = 0, inside = 1;
    while (inside && steps < g.n * sqrt(3)) {
      v3 pos = ray_at(org, t, dir);
      int v = voxel_of(pos, g);
      if (v > (g.n * g.n * g.n - 1) || v < 0) {
        inside = 0;
        st3(pos_out + 3 * (int64_t)i, pos);
        break;
      } else {
        float d = next_voxel_dist(pos, dir, idir, g.n);
        t += d;
        t += eps;
        if (occ[v]) {
          st3(pos_out + 3 * (int64_t)i, pos);
          break;
        }
      }
      steps++;
    }
    within[i] = (uint8_t)inside;
  }
}

/* ------------------------------------------------------------------ packed samples (RaySamplesPackedGPU.cuh) */
/* compact_to_valid_samples_gpu (:15-81) */
void orc_compact(int R, const float* pos, const float* pos4, const float* dirs, const float* z, const float* dt,
                 const float* sdf, const float* fdt, const int* se, float* o_pos, float* o_pos4, float* o_dirs, float* o_z,
                 float* o_dt, float* o_sdf, float* o_fdt, int* o_se, int* o_cur) {
  for (int ray = 0; ray < R; ray++) {
    int s = se[2 * ray], cnt = se[2 * ray + 1] - s;
    int o = o_cur[0];
    o_cur[0] += cnt;
    for (int i = 0; i < cnt; i++) {
      memcpy(o_pos + 3 * (int64_t)(o + i), pos + 3 * (int64_t)(s + i), 12);
      memcpy(o_pos4 + 4 * (int64_t)(o + i), pos4 + 4 * (int64_t)(s + i), 16);
      memcpy(o_dirs + 3 * (int64_t)(o + i), dirs + 3 * (int64_t)(s + i), 12);
      o_z[o + i] = z[s + i];
      o_dt[o + i] = dt[s + i];
      o_sdf[o + i] = sdf[s + i];
    }
    o_fdt[ray] = fdt[ray];
    o_se[2 * ray] = o;
    o_se[2 * ray + 1] = o + cnt;
  }
}
/* compute_per_sample_ray_idx_gpu (:84-115) */
void orc_per_sample_ray_idx(int R, int M, const int* se, int* out) {
  for (int ray = 0; ray < R; ray++)
    for (int i = se[2 * ray]; i < se[2 * ray + 1]; i++)
      if (i < M) out[i] = ray;
}

/* ------------------------------------------------------------------ background sampler (RaySamplerGPU.cuh:37-158) */
void orc_samples_bg(int R, int per_ray, const float* origins, const float* dirs, const float* t_exit_p, float radius,
                    const float* center, uint64_t st, uint64_t inc, int randomize, int contract, float* p3, float* p4,
                    float* s_dirs, float* s_z, float* s_dt, float* fixed_dt, int* start_end) {
  for (int ray = 0; ray < R; ray++) {
    float t_exit = t_exit_p[ray];
    v3 dir = ld3(dirs + 3 * (int64_t)ray), org = ld3(origins + 3 * (int64_t)ray), c = ld3(center);
    float min_t = 1e-3;
    float step = (float)((1.0 - min_t) / (per_ray - 1));
    pcg rng = {st, inc};
    int64_t base = (int64_t)ray * per_ray;
    for (int i = 0; i < per_ray; i++) {
      float ts = (float)(1.0 - i * step);
      if (randomize) {
        pcg_advance(&rng, ray * per_ray);
        float rnd = pcg_next_float(&rng);
        float mov = (float)(step * rnd - step / 2.0);
        ts += mov;
      }
      ts = clampf(ts, min_t, 1.0);
      float zs = t_exit / ts;
      s_z[base + i] = zs;
      v3 p = ray_at(org, zs, dir);
      if (contract) {
        float tr0 = ts * radius;
        float len = sqrtf(dot3(p, p));
        v3 u = mk3(p.x / len, p.y / len, p.z / len);
        p = scale3(2 * radius - tr0, u);
      }
      st3(p3 + 3 * (base + i), p);
      v3 q = sub3(p, c);
      float inv_len = 1.0f / sqrtf(dot3(q, q)); /* rsqrtf on the host */
      v3 u = scale3(inv_len, q);
      float dist = sqrtf(dot3(q, q));
      float* o4 = p4 + 4 * (base + i);
      o4[0] = u.x; o4[1] = u.y; o4[2] = u.z;
      o4[3] = radius / fmaxf(1e-6, dist);
      st3(s_dirs + 3 * (base + i), dir);
    }
    for (int i = 0; i < per_ray - 1; i++) s_dt[base + i] = s_z[base + i + 1] - s_z[base + i];
    s_dt[base + per_ray - 1] = 1e10;
    fixed_dt[ray] = 0;
    start_end[2 * ray] = ray * per_ray;
    start_end[2 * ray + 1] = ray * per_ray + per_ray;
  }
}

/* ------------------------------------------------------------------ sphere (SphereGPU.cuh:21-130) */
void orc_sphere_intersect(int R, float radius, const float* center, const float* origins, const float* dirs, float* p0,
                          float* t0o, float* p1, float* t1o, uint8_t* hit) {
  for (int i = 0; i < R; i++) {
    v3 o = ld3(origins + 3 * (int64_t)i), d = ld3(dirs + 3 * (int64_t)i);
    v3 oc = sub3(o, ld3(center));
    float a = dot3(d, d);
    float b = (float)(2.0 * dot3(oc, d));
    float c = dot3(oc, oc) - radius * radius;
    float disc = b * b - 4 * a * c;
    float t0 = (float)((-b - sqrtf(fabsf(disc))) / (2.0 * a));
    float t1 = (float)((-b + sqrtf(fabsf(disc))) / (2.0 * a));
    int miss = disc < 0;
    if (miss) { t0 = 0.0; t1 = 0.0; }
    t0 = fmaxf(0.0f, t0);
    st3(p0 + 3 * (int64_t)i, ray_at(o, t0, d));
    st3(p1 + 3 * (int64_t)i, ray_at(o, t1, d));
    t0o[i] = t0;
    t1o[i] = t1;
    hit[i] = !miss;
  }
}
void orc_rand_points_inside(int n, float radius, const float* phi, const float* costheta, const float* u, float* out) {
  for (int i = 0; i < n; i++) {
    float theta = acosf(costheta[i]);
    float r = (float)(radius * pow(u[i], 1.0 / 3));
    out[3 * (int64_t)i] = r * sinf(theta) * cosf(phi[i]);
    out[3 * (int64_t)i + 1] = r * sinf(theta) * sinf(phi[i]);
    out[3 * (int64_t)i + 2] = r * cosf(theta);
  }
}

/* ------------------------------------------------------------------ volume rendering (VolumeRenderingGPU.cuh) */
typedef struct { const int* se; int equal, fixed, maxn; } rayidx;
static int ray_range(rayidx ri, int ray, int* s, int* e) {
  if (ri.equal) { *s = ray * ri.fixed; *e = *s + ri.fixed; } else { *s = ri.se[2 * ray]; *e = ri.se[2 * ray + 1]; }
  return !(*e > ri.maxn || (*e - *s) == 0);
}
#define RI const int* se, int equal, int fixed, int maxn
#define MKRI rayidx ri = {se, equal, fixed, maxn}
/* volume_render_nerf (:68-155) */
void orc_volume_render_nerf(int R, RI, const float* rgb, const float* sigma, const float* z, const float* dt, float* pred,
                            float* depth, float* bg, float* w) {
  MKRI;
  for (int ray = 0; ray < R; ray++) {
    int s, e;
    if (!ray_range(ri, ray, &s, &e)) {
      pred[3 * ray] = pred[3 * ray + 1] = pred[3 * ray + 2] = 0;
      depth[ray] = 0;
      bg[ray] = 1.0;
      continue;
    }
    float T = 1.f, r = 0, g = 0, b = 0, dep = 0;
    for (int i = s; i < e; i++) {
      if (T < 1e-4f) break;
      float alpha = 1.f - expf(-sigma[i] * dt[i]);
      float weight = alpha * T;
      r += weight * rgb[3 * (int64_t)i];
      g += weight * rgb[3 * (int64_t)i + 1];
      b += weight * rgb[3 * (int64_t)i + 2];
      dep += weight * z[i];
      T *= (1.f - alpha);
      w[i] = weight;
    }
    pred[3 * ray] = r; pred[3 * ray + 1] = g; pred[3 * ray + 2] = b;
    depth[ray] = dep;
    bg[ray] = T;
  }
}
/* volume_render_nerf_backward (:158-303) */
void orc_volume_render_nerf_backward(int R, RI, const float* g_pred, const float* g_bg, const float* pred, const float* bg,
                                     const float* rgb, const float* sigma, const float* dt, float* g_rgb, float* g_sigma) {
  MKRI;
  for (int ray = 0; ray < R; ray++) {
    int s, e;
    if (!ray_range(ri, ray, &s, &e)) continue;
    float T = 1.f;
    v3 G = ld3(g_pred + 3 * ray), full = ld3(pred + 3 * ray), upto = mk3(0, 0, 0);
    float gbg = g_bg[ray], last_T = bg[ray];
    for (int i = s; i < e; i++) {
      if (T < 1e-4f) break;
      v3 c = ld3(rgb + 3 * (int64_t)i);
      float d = dt[i];
      float alpha = 1.f - expf(-sigma[i] * d);
      float weight = alpha * T;
      upto = add3(upto, scale3(weight, c));
      g_rgb[3 * (int64_t)i] = G.x * weight;
      g_rgb[3 * (int64_t)i + 1] = G.y * weight;
      g_rgb[3 * (int64_t)i + 2] = G.z * weight;
      T *= (1.f - alpha);
      v3 suffix = sub3(full, upto);
      float grad = 0;
      grad += G.x * d * (T * c.x - suffix.x);
      grad += G.y * d * (T * c.y - suffix.y);
      grad += G.z * d * (T * c.z - suffix.z);
      grad += gbg * (-d * last_T);
      g_sigma[i] = grad;
    }
  }
}
/* compute_dt_gpu (:307-367) */
void orc_compute_dt(int R, RI, const float* z, const float* t_exit, int use_t_exit, float* dt) {
  MKRI;
  for (int ray = 0; ray < R; ray++) {
    int s, e;
    if (!ray_range(ri, ray, &s, &e)) continue;
    for (int i = s; i < e; i++) {
      float next = (i < e - 1) ? z[i + 1] : (use_t_exit ? t_exit[ray] : 1e10f);
      dt[i] = next - z[i];
    }
  }
}
/* cumprod_alpha2transmittance_gpu (:371-422) */
void orc_cumprod(int R, RI, const float* alpha, float* T_out, float* bg) {
  MKRI;
  for (int ray = 0; ray < R; ray++) {
    int s, e;
    if (!ray_range(ri, ray, &s, &e)) continue;
    float T = 1.f;
    for (int i = s; i < e; i++) {
      T_out[i] = T;
      if (i < e - 1) T *= alpha[i];
    }
    bg[ray] = T;
  }
}
/* cumprod_alpha2transmittance_backward_gpu (:1135-1205) */
void orc_cumprod_backward(int R, RI, const float* g_bg, const float* alpha, const float* bg, const float* cumsumLV,
                          float* g_alpha) {
  MKRI;
  for (int ray = 0; ray < R; ray++) {
    int s, e;
    if (!ray_range(ri, ray, &s, &e)) continue;
    for (int i = s; i < e; i++) {
      float g = 0;
      if (i < e - 1) {
        g = cumsumLV[i + 1] / fmaxf(1e-6, alpha[i]);
        g += g_bg[ray] * bg[ray] / fmaxf(1e-6, alpha[i]);
      }
      g_alpha[i] = g;
    }
  }
}
/* integrate_with_weights_gpu (:425-481) */
void orc_integrate(int R, RI, const float* rgb, const float* w, float* pred) {
  MKRI;
  for (int ray = 0; ray < R; ray++) {
    int s, e;
    if (!ray_range(ri, ray, &s, &e)) continue;
    v3 acc = mk3(0, 0, 0);
    for (int i = s; i < e; i++) acc = add3(acc, scale3(w[i], ld3(rgb + 3 * (int64_t)i)));
    st3(pred + 3 * ray, acc);
  }
}
/* integrate_with_weights_backward_gpu (:1208-1269); compat=1 keeps the [1]-for-[2] channel read of :1247 */
void orc_integrate_backward(int R, RI, const float* g_pred, const float* rgb, const float* w, float* g_rgb, float* g_w,
                            int compat) {
  MKRI;
  for (int ray = 0; ray < R; ray++) {
    int s, e;
    if (!ray_range(ri, ray, &s, &e)) continue;
    v3 G = ld3(g_pred + 3 * ray);
    for (int i = s; i < e; i++) {
      v3 c = ld3(rgb + 3 * (int64_t)i);
      if (compat) c.z = c.y;
      g_rgb[3 * (int64_t)i] = G.x * w[i];
      g_rgb[3 * (int64_t)i + 1] = G.y * w[i];
      g_rgb[3 * (int64_t)i + 2] = G.z * w[i];
      g_w[i] = G.x * c.x + G.y * c.y + G.z * c.z;
    }
  }
}
static float map_range(float v, float i0, float i1, float o0, float o1) {
  float c = fmaxf(i0, fminf(i1, v));
  return o0 + ((o1 - o0) / (i1 - i0)) * (c - i0);
}
static float sigmoid_(float x) { return (float)(1.0 / (1.0 + expf(-x))); }
/* sdf2alpha_gpu (:490-564) */
void orc_sdf2alpha(int R, RI, const float* fixed_dt, const float* dt, const float* sdf, float inv_s, int dynamic, float mult,
                   float* alpha) {
  MKRI;
  for (int ray = 0; ray < R; ray++) {
    float s_ = inv_s;
    if (dynamic) s_ = map_range(fixed_dt[ray], 0.0001, 0.01, 1024, 64);
    s_ = s_ * mult;
    int s, e;
    if (!ray_range(ri, ray, &s, &e)) continue;
    for (int i = s; i < e - 1; i++) {
      float d = dt[i], prev = sdf[i], next = sdf[i + 1];
      float mid = (float)((prev + next) * 0.5);
      float cosv = (next - prev) / fmaxf(1e-6, d);
      cosv = clampf(cosv, -1e3, 0.0);
      float prev_e = (float)(mid - cosv * d * 0.5);
      float next_e = (float)(mid + cosv * d * 0.5);
      float pc = sigmoid_(prev_e * s_), nc = sigmoid_(next_e * s_);
      alpha[i] = (float)((pc - nc + 1e-6) / (pc + 1e-6));
    }
  }
}
/* sum_over_each_ray_gpu (:566-628) */
void orc_sum_over_each_ray(int R, RI, int C, const float* v, float* s_ray, float* s_smp) {
  MKRI;
  for (int ray = 0; ray < R; ray++) {
    int s, e;
    if (!ray_range(ri, ray, &s, &e)) continue;
    for (int c = 0; c < C; c++) {
      float acc = 0;
      for (int i = s; i < e; i++) acc += v[(int64_t)i * C + c];
      s_ray[ray * C + c] = acc;
      for (int i = s; i < e; i++) s_smp[(int64_t)i * C + c] = acc;
    }
  }
}
/* sum_over_each_ray_backward_gpu (:1271-1329) */
void orc_sum_over_each_ray_backward(int R, RI, int C, const float* g_ray, const float* g_smp, float* g) {
  MKRI;
  for (int ray = 0; ray < R; ray++) {
    int s, e;
    if (!ray_range(ri, ray, &s, &e)) continue;
    for (int i = s; i < e; i++)
      for (int c = 0; c < C; c++) g[(int64_t)i * C + c] = g_ray[ray * C + c] + g_smp[(int64_t)i * C + c];
  }
}
/* cumsum_over_each_ray_gpu (:631-691) and compute_cdf_gpu (:697-752, exclusive=1) */
void orc_cumsum(int R, RI, const float* v, int inverse, int exclusive, float* out) {
  MKRI;
  for (int ray = 0; ray < R; ray++) {
    int s, e;
    if (!ray_range(ri, ray, &s, &e)) continue;
    float acc = 0;
    for (int i = 0; i < e - s; i++) {
      int idx = inverse ? (e - 1 - i) : (s + i);
      if (exclusive) { out[idx] = acc; acc += v[idx]; } else { acc += v[idx]; out[idx] = acc; }
    }
  }
}
/* binary_search (:764-789) */
static int cdf_search(const float* cdf, float val, int imin, int imax) {
  if (imax - imin < 1) return imax; /* the reference does not terminate on a single-sample ray */
  while (imax >= imin) {
    int imid = imin + (imax - imin) / 2;
    if (cdf[imid] > val) imax = imid; else imin = imid;
    if ((imax - imin) == 1) return imax;
  }
  return imax;
}
/* importance_sample_gpu (:793-946) */
void orc_importance_sample(int R, RI, const float* origins, const float* dirs, const float* fixed_dt_p, const float* z,
                           const float* cdf, int nimp, uint64_t st, uint64_t inc, int jitter, float* o_pos, float* o_dirs,
                           float* o_z) {
  MKRI;
  for (int ray = 0; ray < R; ray++) {
    int s, e;
    int ok = ray_range(ri, ray, &s, &e);
    int64_t ob = (int64_t)ray * nimp;
    if (!ok) {
      for (int i = 0; i < nimp; i++) {
        st3(o_pos + 3 * (ob + i), mk3(0, 0, 0));
        st3(o_dirs + 3 * (ob + i), mk3(0, 0, 0));
        o_z[ob + i] = -1;
      }
      continue;
    }
    v3 org = ld3(origins + 3 * (int64_t)ray), dir = ld3(dirs + 3 * (int64_t)ray);
    float fixed_dt = fixed_dt_p[ray];
    pcg rng = {st, inc};
    for (int i = 0; i < nimp; i++) {
      float step = (float)(1.0 / (nimp + 1));
      float u = step + i * step;
      if (jitter) {
        pcg_advance(&rng, ray);
        float rnd = pcg_next_float(&rng);
        float mov = (float)(step / 2.0);
        u += map_range(rnd, 0.0, 1.0, -mov, +mov);
      }
      u = clampf(u, 0.0 + 1e-6, 1.0 - 1e-5);
      int imax = cdf_search(cdf, u, s, e - 1);
      int imin = imax - 1 > 0 ? imax - 1 : 0;
      float cdf_max = cdf[imax], cdf_min = cdf[imin];
      float z_max = z[imax], z_min = z[imin];
      float z_imp = map_range(u, cdf_min, cdf_max, z_min, z_max);
      float d_min = z_imp - z_min, d_max = z_max - z_imp;
      if (d_min < d_max) {
        d_min = fminf(d_min, fixed_dt);
        z_imp = z_min + d_min;
      } else {
        d_max = fminf(d_max, fixed_dt);
        z_imp = z_max - d_max;
      }
      st3(o_pos + 3 * (ob + i), ray_at(org, z_imp, dir));
      st3(o_dirs + 3 * (ob + i), dir);
      o_z[ob + i] = z_imp;
    }
  }
}
/* combine_uniform_samples_with_imp_gpu (:950-1131) */
void orc_combine(int R, RI, const float* origins, const float* dirs, const float* t_exit, const float* uni_fdt,
                 const float* uni_z, const float* uni_sdf, int has_sdf, int nimp, const float* imp_z, const float* imp_sdf,
                 int out_max, float* c_pos, float* c_dirs, float* c_z, float* c_dt, float* c_sdf, float* c_fdt, int* c_se,
                 int* c_cur) {
  MKRI;
  for (int ray = 0; ray < R; ray++) {
    int us, ue;
    ray_range(ri, ray, &us, &ue);
    int un = ue - us;
    if (un <= 1) {
      c_fdt[ray] = 0;
      c_se[2 * ray] = 0;
      c_se[2 * ray + 1] = 0;
      continue;
    }
    int total = un + nimp;
    int base = c_cur[0];
    c_cur[0] += total;
    c_se[2 * ray] = base;
    c_se[2 * ray + 1] = base + total;
    if (base + total > out_max) continue;
    v3 org = ld3(origins + 3 * (int64_t)ray), dir = ld3(dirs + 3 * (int64_t)ray);
    float fixed_dt = uni_fdt[ray];
    c_fdt[ray] = fixed_dt;
    int is = ray * nimp, cu = 0, ci = 0;
    for (int i = 0; i < total; i++) {
      float zu = cu < un ? uni_z[us + cu] : 1e10f;
      float zi = ci < nimp ? imp_z[is + ci] : 1e10f;
      int take_u = zu < zi;
      float zz = take_u ? zu : zi;
      int64_t o = base + i;
      st3(c_pos + 3 * o, ray_at(org, zz, dir));
      st3(c_dirs + 3 * o, dir);
      c_z[o] = zz;
      if (has_sdf) c_sdf[o] = take_u ? uni_sdf[us + cu] : imp_sdf[is + ci];
      if (take_u) cu++; else ci++;
    }
    for (int i = 0; i < total - 1; i++) c_dt[base + i] = fminf(c_z[base + i + 1] - c_z[base + i], fixed_dt);
    c_dt[base + total - 1] = clampf(t_exit[ray] - c_z[base + total - 1], 0.0, fixed_dt);
  }
}

/* ------------------------------------------------------------------ spherical harmonics (PermutoSDFGPU.cuh:275-365) */
void orc_spherical_harmonics(int n, int degree, const float* dirs, float* out) {
  const int ch = degree * degree;
  for (int i = 0; i < n; i++) {
    float x = dirs[3 * (int64_t)i], y = dirs[3 * (int64_t)i + 1], z = dirs[3 * (int64_t)i + 2];
    float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    float x4 = x2 * x2, y4 = y2 * y2, z4 = z2 * z2, x6 = x4 * x2, y6 = y4 * y2, z6 = z4 * z2;
    float* o = out + (int64_t)i * ch;
    o[0] = 0.28209479177387814f;
    if (degree <= 1) continue;
    o[1] = -0.48860251190291987f * y; o[2] = 0.48860251190291987f * z; o[3] = -0.48860251190291987f * x;
    if (degree <= 2) continue;
    o[4] = 1.0925484305920792f * xy; o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz; o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    if (degree <= 3) continue;
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2); o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2); o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2); o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    if (degree <= 4) continue;
    o[16] = 2.5033429417967046f * xy * (x2 - y2); o[17] = 1.7701307697799304f * yz * (-3.0f * x2 + y2);
    o[18] = 0.94617469575756008f * xy * (7.0f * z2 - 1.0f); o[19] = 0.66904654355728921f * yz * (3.0f - 7.0f * z2);
    o[20] = -3.1735664074561294f * z2 + 3.7024941420321507f * z4 + 0.31735664074561293f;
    o[21] = 0.66904654355728921f * xz * (3.0f - 7.0f * z2); o[22] = 0.47308734787878004f * (x2 - y2) * (7.0f * z2 - 1.0f);
    o[23] = 1.7701307697799304f * xz * (-x2 + 3.0f * y2);
    o[24] = -3.7550144126950569f * x2 * y2 + 0.62583573544917614f * x4 + 0.62583573544917614f * y4;
    if (degree <= 5) continue;
    o[25] = 0.65638205684017015f * y * (10.0f * x2 * y2 - 5.0f * x4 - y4); o[26] = 8.3026492595241645f * xy * z * (x2 - y2);
    o[27] = -0.48923829943525038f * y * (3.0f * x2 - y2) * (9.0f * z2 - 1.0f);
    o[28] = 4.7935367849733241f * xy * z * (3.0f * z2 - 1.0f);
    o[29] = 0.45294665119569694f * y * (14.0f * z2 - 21.0f * z4 - 1.0f);
    o[30] = 0.1169503224534236f * z * (-70.0f * z2 + 63.0f * z4 + 15.0f);
    o[31] = 0.45294665119569694f * x * (14.0f * z2 - 21.0f * z4 - 1.0f);
    o[32] = 2.3967683924866621f * z * (x2 - y2) * (3.0f * z2 - 1.0f);
    o[33] = -0.48923829943525038f * x * (x2 - 3.0f * y2) * (9.0f * z2 - 1.0f);
    o[34] = 2.0756623148810411f * z * (-6.0f * x2 * y2 + x4 + y4);
    o[35] = 0.65638205684017015f * x * (10.0f * x2 * y2 - x4 - 5.0f * y4);
    if (degree <= 6) continue;
    o[36] = 1.3663682103838286f * xy * (-10.0f * x2 * y2 + 3.0f * x4 + 3.0f * y4);
    o[37] = 2.3666191622317521f * yz * (10.0f * x2 * y2 - 5.0f * x4 - y4);
    o[38] = 2.0182596029148963f * xy * (x2 - y2) * (11.0f * z2 - 1.0f);
    o[39] = -0.92120525951492349f * yz * (3.0f * x2 - y2) * (11.0f * z2 - 3.0f);
    o[40] = 0.92120525951492349f * xy * (-18.0f * z2 + 33.0f * z4 + 1.0f);
    o[41] = 0.58262136251873131f * yz * (30.0f * z2 - 33.0f * z4 - 5.0f);
    o[42] = 6.6747662381009842f * z2 - 20.024298714302954f * z4 + 14.684485723822165f * z6 - 0.31784601133814211f;
    o[43] = 0.58262136251873131f * xz * (30.0f * z2 - 33.0f * z4 - 5.0f);
    o[44] = 0.46060262975746175f * (x2 - y2) * (11.0f * z2 * (3.0f * z2 - 1.0f) - 7.0f * z2 + 1.0f);
    o[45] = -0.92120525951492349f * xz * (x2 - 3.0f * y2) * (11.0f * z2 - 3.0f);
    o[46] = 0.50456490072872406f * (11.0f * z2 - 1.0f) * (-6.0f * x2 * y2 + x4 + y4);
    o[47] = 2.3666191622317521f * xz * (10.0f * x2 * y2 - x4 - 5.0f * y4);
    o[48] = 10.247761577878714f * x2 * y4 - 10.247761577878714f * x4 * y2 + 0.6831841051919143f * x6 - 0.6831841051919143f * y6;
  }
}

/* ------------------------------------------------------------------ rays from the image reel (PermutoSDFGPU.cuh:24-127) */
void orc_random_rays_from_reel(int R, int H, int W, const float* rgb, const float* mask, const float* K, const float* tf,
                               const int* pix, const int* img, int has_mask, float* o, float* d, float* gt, float* gm) {
  for (int i = 0; i < R; i++) {
    int im = img[i], p = pix[i];
    float px = (float)(p % W), py = (float)(p / W);
    px = (float)(px + 0.5);
    py = (float)(py + 0.5);
    const float* Ki = K + 9 * (int64_t)im;
    float fx = Ki[0], fy = Ki[4], cx = Ki[2], cy = Ki[5];
    v3 pc = mk3((px - cx) / fx, (py - cy) / fy, 1.0);
    const float* T = tf + 16 * (int64_t)im;
    v3 t = mk3(T[3], T[7], T[11]);
    v3 pw = mk3(T[0] * pc.x + T[1] * pc.y + T[2] * pc.z, T[4] * pc.x + T[5] * pc.y + T[6] * pc.z,
                T[8] * pc.x + T[9] * pc.y + T[10] * pc.z);
    pw = add3(pw, t);
    v3 dd = sub3(pw, t);
    float inv = 1.0f / sqrtf(dot3(dd, dd));
    v3 dir = scale3(inv, dd);
    int x = (int)floor(px), y = (int)floor(py);
    int64_t plane = (int64_t)H * W;
    const float* c = rgb + (int64_t)im * 3 * plane + (int64_t)y * W + x;
    float m = has_mask ? mask[(int64_t)im * plane + (int64_t)y * W + x] : 1.0f;
    st3(o + 3 * (int64_t)i, t);
    st3(d + 3 * (int64_t)i, dir);
    gt[3 * (int64_t)i] = c[0] * m;
    gt[3 * (int64_t)i + 1] = c[plane] * m;
    gt[3 * (int64_t)i + 2] = c[2 * plane] * m;
    gm[i] = m;
  }
}
