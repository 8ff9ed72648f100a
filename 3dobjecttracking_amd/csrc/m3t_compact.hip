// m3t_compact.hip — tracking_step_compact_kernel: Tracker::ExecuteTrackingStep (tracker.cpp:344-364) for batches
// that fill the chip (two objects per CU and more).  Included by m3t_hip_api.hip after m3t_kernels.hip, whose
// device functions (view search, occlusion windows, depth scan, solve, histogram update) it shares.
//
// tracking_step_kernel keeps ~75 KB of LDS and > 200 VGPRs per object: two workgroups per CU, and a lone wave of
// this chip issues one VALU instruction per ~4.7 cycles (6.75 when dependent, tools/ubench.hip), so a CU with two
// waves per SIMD spends most cycles waiting.  Here one object needs <= 30 KB (Region) / <= 47 KB (Region + Depth)
// and <= 128 VGPRs: four to five 256-thread workgroups per CU, 16-20 resident waves.
//   * correspondence search: ONE thread walks a whole correspondence line (CalculateSegmentProbabilities
//     region_modality.cpp:1433-1573 in walk order, the v_f += v_step chain continuing in a register), keeps the last
//     eight segment probabilities in a register ring and finishes one value of CalculateDistribution (:1600-1637)
//     per segment from it: no chain / segment buffers, no barrier between line set-up, pixel walk, distribution and
//     moments.  Lines that fill their segments from the far end (negative dominant normal component) read the
//     ring newest-first, the others oldest-first: the products keep the reference's order for both.
//   * gradient / Hessian: a line stores its eight factors (J[6], weight * dll, weight / variance) instead of 27
//     products; the 42 chain lanes form their product from three rows -- (a * b) * c, the reference's expression --
//     right before the dependent subtraction (same operations, same order: bit-identical sums).
// Requires function_length 8, distribution_length 12 (the reference's defaults, used by every configuration of
// SURVEY Appendix C), scales <= 9, n_lines_max <= 256; the host falls back to tracking_step_kernel otherwise.

namespace {

constexpr int kCMiscPose = 544, kCMiscGhR = 560, kCMiscGhD = 608;  // inside the M3T_COMPACT_MISC_FLOATS block
static_assert(kMiscLogTable + 2 * M3T_LOG_TABLE_DOUBLES <= kCMiscPose, "log table overlaps the pose");
static_assert(kCMiscGhD + 42 <= M3T_COMPACT_MISC_FLOATS, "misc block too small");

// One correspondence line, walked by one thread.  dist0 = &state[CS_DIST0 * nl + line]: receives the 12 raw
// distribution products (CalculateDistribution before its normalisation).
// The pair table of a 32-bin histogram is 256 KB: every sample of the walk is a dependent 8-byte gather from L2
// (98.8 k requests per object-frame: with the pixel loads they keep the texture-address units 66 % busy,
// profiles/r05_pmc_rbot4096.json).  But a pair takes one of three constants for almost every bin -- (0.5, 0.5) for a
// bin neither histogram holds (region_modality.cpp:1594-1597), (1, 0) / (0, 1) for a bin only one of them holds
// (x / (x + 0) == 1 exactly) -- and only the bins BOTH hold (a few hundred of 32 768) need their two floats.  So the
// table is kept in LDS as two bits per bin plus the pairs of the mixed bins in bin order:
//   word w: a = bits of the bins whose pair is not (0.5, 0.5) / (0, 1), b = ... not (0.5, 0.5) / (1, 0); a & b = mixed
//   rank[w] = number of mixed bins in front of word w;  pairs[rank] = the pair as the blend stored it
// classified from the stored PAIR (not from the histograms), so the value a lookup returns is the table's value bit for
// bit whatever produced it.  A mixed bin whose rank does not fit the LDS budget is read from the global table.
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
// A stored pair takes FIVE bytes, not eight: the smaller of its two floats as it is, and the larger one as its distance in
// units in the last place from 1.0f - smaller (x + y = 1 up to roundings: the distance is -2 .. 2), with a bit that says
// which of the two the smaller one is.  larger = as_float(as_int(1.0f - smaller) + k) is the stored float bit for bit --
// k is the difference of the two bit patterns, nothing is assumed about the roundings; a pair whose k does not fit four
// bits (never seen) is marked and read from the global table.  1.6 x the pairs in the same LDS: the bench's steady state
// (2 000 mixed bins after a few hundred frames of the same scene) fits where 1 400 eight-byte pairs did not.
struct CompactTable {  // (address-space-3 pointers: ds_read, not flat loads)
  const __attribute__((address_space(3))) v2u* ab;
  const __attribute__((address_space(3))) uint16_t* rank;  // 3 + the number of mixed bins in front of the word
  const __attribute__((address_space(3))) float* small;    // [3 + cap]: the smaller float of the pair in slot i
  const __attribute__((address_space(3))) uint8_t* meta;   // [3 + cap]: bit 7 = y is the smaller one, bits 0-3 = k + 8
  int cap;
  bool fits;  // every mixed bin of this object has its slot (else the workgroup walks with the global table)
};
__device__ __forceinline__ uint8_t compact_pair_meta(float x, float y, float* smaller) {
  const bool y_small = y < x;
  const float s = y_small ? y : x, l = y_small ? x : y;
  const int k = __float_as_int(l) - __float_as_int(1.0f - s);
  *smaller = s;
  if (k < -8 || k > 7) return 0xffu;
  return (uint8_t)((y_small ? 0x80 : 0) | (k + 8));
}

template <int SCALE, bool TABLE = false>
__device__ __forceinline__ void compact_walk(G<uint8_t> image, uint32_t pitch, G<v2f> hist, int bitshift, int bin_bits,
                                             int start, float step, float x0, bool horiz, bool reversed,
                                             const float (&lf)[8], const float (&lb)[8], float* dist0, int nl,
                                             const CompactTable* table = nullptr) {
  // segments per batch: all pixel loads of a batch are issued before the first histogram gather
  // 8 - 20 pixel loads in flight per lane: measured best.  (Smaller batches, and smaller batches with the next
  // batch's pixels requested before the current one is worked on, were both slower: 2.02 / 1.99 vs 1.95 ms per
  // 4096-object step -- with 16 waves per CU the other waves are the prefetch.)
#ifndef M3T_TABLE_GS_SHIFT
#define M3T_TABLE_GS_SHIFT 1  /* the table walk's batches: half the plain walk's (4096 objects: 1.88 / 1.78 / 1.85 ms for 0 / 1 / 2) */
#endif
#ifndef M3T_TABLE_WAVES
#define M3T_TABLE_WAVES 3     /* experiment knob: waves per SIMD the table kernel's register budget is held to */
#endif
  constexpr int GS0 = SCALE <= 2 ? 8 : (SCALE <= 5 ? 4 : 2);
  constexpr int GS = TABLE ? ((GS0 >> M3T_TABLE_GS_SHIFT) > 0 ? (GS0 >> M3T_TABLE_GS_SHIFT) : 1) : GS0;
  float wf[8], wb[8];  // ring: segment t of the walk sits in slot t & 7
#pragma unroll
  for (int i = 0; i < 8; ++i) { wf[i] = 0.0f; wb[i] = 0.0f; }
  float x = x0;
  const unsigned long long rev_mask = __builtin_amdgcn_ballot_w64(reversed);  // lanes that fill from the far end
  // byte offset = minor * stride_minor + major * stride_major; minor < 2^16 and the strides < 2^24: the 24-bit
  // multiply is exact in its low 32 bits
  const uint32_t stride_minor = horiz ? pitch : 3u, stride_major = horiz ? 3u : pitch;
  uint32_t off_major = (uint32_t)start * stride_major;
  // pixel loads of the batch that starts at walk segment `first` (the v_f += v_step chain runs on through the batches)
  auto load_batch = [&](uint32_t (&px)[GS][SCALE], int first) {
#pragma unroll
    for (int g = 0; g < GS; ++g) {
      const bool live = first + g < 19;
#pragma unroll
      for (int j = 0; j < SCALE; ++j) {
        px[g][j] = 0;
        if (live) {
          const uint32_t off = __umul24((uint32_t)f2i(x), stride_minor) + off_major;
          px[g][j] = reinterpret_cast<G<PackedU32>>(image + off)->v;
          off_major += stride_major;
          x += step;  // the reference's v_f += v_step chain (:1464-1473), one rounding per pixel
        }
      }
    }
  };
#pragma nounroll
  for (int t0 = 0; t0 < 19; t0 += 8) {
#pragma unroll
    for (int g0 = 0; g0 < 8; g0 += GS) {
      if (t0 + g0 < 19) {  // uniform (no `break`: it would send the unrolled loop's arrays to scratch memory)
      uint32_t px[GS][SCALE];
      load_batch(px, t0 + g0);
      v2f h[GS][SCALE];
      if constexpr (!TABLE) {
#pragma unroll
      for (int g = 0; g < GS; ++g)
#pragma unroll
        for (int j = 0; j < SCALE; ++j) {
          const uint32_t v = px[g][j];
          // (B >> s) * n^2 + (G >> s) * n + (R >> s) with n = 2^bin_bits (color_histograms.cpp:97-99)
          const uint32_t idx = ((((v & 0xffu) >> bitshift) << bin_bits | ((v >> 8) & 0xffu) >> bitshift) << bin_bits) |
                               (((v >> 16) & 0xffu) >> bitshift);
          h[g][j] = hist[idx];
        }
      } else {
        // all words of the batch first, then ONE pair per sample -- slot 0 / 1 / 2 of the pair array hold the three
        // constants, a mixed bin's pair sits at 3 + its rank: the slot is a select, not the value.  (Every mixed bin of
        // this object has its slot: a workgroup whose table does not hold them all walks with the global table.)
        v2u ab[GS][SCALE];
        uint32_t rk[GS][SCALE];
#pragma unroll
        for (int g = 0; g < GS; ++g)
#pragma unroll
          for (int j = 0; j < SCALE; ++j) {
            const uint32_t v = px[g][j];
            const uint32_t idx = ((((v & 0xffu) >> bitshift) << bin_bits | ((v >> 8) & 0xffu) >> bitshift) << bin_bits) |
                                 (((v >> 16) & 0xffu) >> bitshift);
            px[g][j] = idx;
            ab[g][j] = table->ab[idx >> 5];
            rk[g][j] = table->rank[idx >> 5];  // 3 + the mixed bins in front of the word
          }
        // the slot of every sample (its bit words and rank are dead after that) ...
#pragma unroll
        for (int g = 0; g < GS; ++g)
#pragma unroll
          for (int j = 0; j < SCALE; ++j) {
            const uint32_t bit = px[g][j] & 31u, a = (ab[g][j].x >> bit) & 1u, b = (ab[g][j].y >> bit) & 1u;
            const uint32_t r = rk[g][j] + (uint32_t)__builtin_popcount((ab[g][j].x & ab[g][j].y) & ((1u << bit) - 1u));
            const uint32_t state = a + 2u * b;  // 0: (0.5, 0.5), 1: (1, 0), 2: (0, 1), 3: mixed
            rk[g][j] = state == 3u ? r : state;
          }
        // ... the stored halves of all pairs of the batch ...
#pragma unroll
        for (int g = 0; g < GS; ++g)
#pragma unroll
          for (int j = 0; j < SCALE; ++j) {
            ab[g][j].x = __float_as_uint(table->small[rk[g][j]]);
            ab[g][j].y = table->meta[rk[g][j]];
          }
        // ... and the pairs
#pragma unroll
        for (int g = 0; g < GS; ++g)
#pragma unroll
          for (int j = 0; j < SCALE; ++j) {
            const float sm = __uint_as_float(ab[g][j].x);
            const uint32_t meta = ab[g][j].y;
            const float lg = __int_as_float(__float_as_int(1.0f - sm) + (int)(meta & 15u) - 8);
            v2f p;
            p.x = (meta & 0x80u) ? lg : sm;
            p.y = (meta & 0x80u) ? sm : lg;
            h[g][j] = p;
          }
      }
#pragma unroll
      for (int g = 0; g < GS; ++g) {
        const int t = t0 + g0 + g;
        if (t < 19) {  // uniform
          float pf = 1.0f, pb = 1.0f;
#pragma unroll
          for (int j = 0; j < SCALE; ++j) {  // MultiplyPixelColorProbability :1575-1598 in walk order
            pf *= h[g][j].x;
            pb *= h[g][j].y;
          }
          if (SCALE > 1) {  // per-segment renormalisation :1556-1571
            if (pf || pb) {
              float sum = pf;
              sum += pb;
              pf /= sum;
              pb /= sum;
            } else {
              pf = 0.5f;
              pb = 0.5f;
            }
          }
          const int s = (g0 + g) & 7;  // static
          wf[s] = pf;
          wb[s] = pb;
          if (t >= 7) {
            // CalculateDistribution :1600-1637: value = prod_k (sf[d + k] * lookup_f[k] + sb[d + k] * lookup_b[k]).
            // Forward fill: segment index == walk index, d = t - 7, term k is walk segment t - 7 + k (oldest first).
            // Reversed fill: segment index == 18 - walk index, d = 18 - t, term k is walk segment t - k (newest first).
            float value = 1.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              // one v_cndmask each, spelled out: left to itself the compiler selects the ring INDEX per lane and
              // emulates the indexed register read with a seven-deep compare / select chain
              float sf, sb;
              asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(sf) : "v"(wf[(s + 1 + k) & 7]), "v"(wf[(s - k) & 7]), "s"(rev_mask));
              asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(sb) : "v"(wb[(s + 1 + k) & 7]), "v"(wb[(s - k) & 7]), "s"(rev_mask));
              value *= sf * lf[k] + sb * lb[k];
            }
            const int d = reversed ? 18 - t : t - 7;
            dist0[d * nl] = value;
          }
        }
      }
      }
    }
  }
}

// Builds the LDS table above from the modality's pair table (histogram_norm) and its occupancy bytes (one per group of
// four bins, 0 = all four pairs are (0.5, 0.5)); 256 threads, thread t takes the words 4 t .. 4 t + 3 of 1024 (32 bins
// each), so that the ranks follow from a scan over the threads.  Ends with a barrier.
__device__ __forceinline__ bool compact_stage_table(CRegion& m, float* lds_table, int n_words, int cap, float* misc,
                                                    unsigned* overflow_word) {
  const int tid = threadIdx.x, nt = blockDim.x;
  v2u* ab = reinterpret_cast<v2u*>(lds_table);
  float* small = lds_table + 2 * n_words;                                                  // [3 + cap]
  uint8_t* meta = reinterpret_cast<uint8_t*>(small + 3 + cap);                             // [3 + cap]
  uint16_t* rank = reinterpret_cast<uint16_t*>(small + 3 + cap + (3 + cap + 3) / 4);       // [n_words]
  if (tid < 3) {  // slot 0: (0.5, 0.5), 1: (1, 0), 2: (0, 1)
    float sm;
    meta[tid] = compact_pair_meta(tid == 0 ? 0.5f : (tid == 1 ? 1.0f : 0.0f), tid == 0 ? 0.5f : (tid == 1 ? 0.0f : 1.0f), &sm);
    small[tid] = sm;
  }
  G<v4f> norm4 = (G<v4f>)m.histogram_norm;          // two pairs per v4f
  G<uint8_t> occupancy = as_global(m.occupancy);    // [n_bins3 / 4]
  const int per_thread = (n_words + nt - 1) / nt;   // 4 for 32 bins, 1 (half the threads idle) for 16
  const int w0 = tid * per_thread;
  int my_mixed = 0;
  for (int k = 0; k < per_thread; ++k) {
    const int w = w0 + k;
    if (w >= n_words) break;
    uint32_t a = 0u, b = 0u;
    const v2u occ = *(G<v2u>)(occupancy + 8 * w);  // the word's eight groups
    const unsigned long long occ64 = ((unsigned long long)occ.y << 32) | occ.x;
    if (occ64 != 0ull) {
      // the pairs of all occupied groups are requested before the first is looked at: one round trip per word
      v4f n[8][2];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if ((occ64 >> (8 * g)) & 0xffull) {
          n[g][0] = norm4[(32 * w + 4 * g) / 2];
          n[g][1] = norm4[(32 * w + 4 * g) / 2 + 1];
        } else {
          n[g][0] = v4f{0.5f, 0.5f, 0.5f, 0.5f};
          n[g][1] = v4f{0.5f, 0.5f, 0.5f, 0.5f};
        }
      }
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const float px[4] = {n[g][0].x, n[g][0].z, n[g][1].x, n[g][1].z}, py[4] = {n[g][0].y, n[g][0].w, n[g][1].y, n[g][1].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool empty = px[i] == 0.5f && py[i] == 0.5f;
          const bool f_only = px[i] == 1.0f && py[i] == 0.0f, b_only = px[i] == 0.0f && py[i] == 1.0f;
          if (!empty && !b_only) a |= 1u << (4 * g + i);
          if (!empty && !f_only) b |= 1u << (4 * g + i);
        }
      }
    }
    v2u e_ab;
    e_ab.x = a;
    e_ab.y = b;
    ab[w] = e_ab;
    my_mixed += __builtin_popcount(a & b);
  }
  // exclusive scan of my_mixed over the threads: inside the wave by shuffles, across the waves through `misc`
  int incl = my_mixed;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int up = __shfl_up(incl, d, kWave);
    if ((tid & (kWave - 1)) >= d) incl += up;
  }
  int* wsum = reinterpret_cast<int*>(misc) + 64;
  if ((tid & (kWave - 1)) == kWave - 1) wsum[tid / kWave] = incl;
  __syncthreads();
  int base = incl - my_mixed;
  for (int wv = 0; wv < tid / kWave; ++wv) base += wsum[wv];
  // the mixed bins' numbers into their slots first (LDS only: a colour cluster puts a hundred of them into one thread's
  // words) ...
  uint32_t* slot_bin = reinterpret_cast<uint32_t*>(small);
  for (int k = 0; k < per_thread; ++k) {
    const int w = w0 + k;
    if (w >= n_words) break;
    rank[w] = (uint16_t)(base + 3);
    const v2u e = ab[w];
    uint32_t mixed = e.x & e.y;
    while (mixed) {
      const int bit = __builtin_ctz(mixed);
      mixed &= mixed - 1u;
      if (base < cap) slot_bin[3 + base] = (uint32_t)(32 * w + bit);
      ++base;
    }
  }
  if (tid == nt - 1) wsum[8] = base;  // all mixed bins of the table
  __syncthreads();
  // ... then the pairs themselves, dealt out over all threads, four requests in flight each
  bool bad_k = false;
  {
    const int n_stored = min(wsum[8], cap);
    G<v2f> norm = (G<v2f>)m.histogram_norm;
    for (int e0 = tid; e0 < n_stored; e0 += 4 * nt) {
      v2f pr[4];
      // (bad_k below: a pair whose larger float is further than the four bits of k reach from 1 - smaller -- never seen)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (e0 + q * nt < n_stored) pr[q] = norm[slot_bin[3 + e0 + q * nt]];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (e0 + q * nt < n_stored) {
          float sm;
          const uint8_t mt = compact_pair_meta(pr[q].x, pr[q].y, &sm);
          bad_k = bad_k || mt == 0xffu;
          meta[3 + e0 + q * nt] = mt;
          small[3 + e0 + q * nt] = sm;
        }
    }
  }
  // histograms whose mixed bins outgrow the table (long sequences): this workgroup walks with the global table, and the
  // host is told by how much, so that it can go back to the kernel without the table (no traffic while everything fits)
  const int total_mixed = wsum[8];
  if (tid == nt - 1 && total_mixed > cap && overflow_word)
    __hip_atomic_fetch_max(overflow_word, (unsigned)(total_mixed - cap), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const bool any_bad = __syncthreads_or(bad_k ? 1 : 0) != 0;
  return total_mixed <= cap && !any_bad;
}

// RegionModality::CalculateCorrespondences (:390-465) for one object by one 256-thread workgroup, thread = line.
// Line set-up as in region_correspondences (phase A), then the walk above, normalisation and moments; everything a
// line needs in between stays in the thread's registers.  Ends with a barrier.  Returns the closest view.
template <bool TABLE = false>
__device__ __forceinline__ int compact_region_correspondences(CRegion& m, CCam& cam, CCam* dcam, const Affine& b2c,
                                                              const Affine& b2dc, int iteration, int corr_iteration,
                                                              float* misc, float* state, int nl, int prev_view = -1,
                                                              const CompactTable* table = nullptr) {
  const int tid = threadIdx.x;
  const RegionIter it = region_iter(m, corr_iteration);
  // the view of the previous search first (closest_view_local: 20 loads, no barrier), else the scan over all views
  int view = -1;
  if (prev_view >= 0 && m.view_neighbors != nullptr) {
    float o0, o1, o2;
    if (view_direction(b2c, o0, o1, o2)) view = closest_view_local((G<v4f>)m.view_neighbors, prev_view, o0, o1, o2);
  }
  if (view < 0) view = closest_view((G<v4f>)m.orientations4, m.n_views, b2c, misc);
  const int n_lines =
      number_of_lines(m.n_lines_max, m.use_adaptive_coverage, m.reference_contour_length, as_global(m.extents), view,
                      m.max_extent, m.n_points);
  const bool handle_occlusions = (iteration - m.first_iteration) >= m.n_unoccluded_iterations;
  const bool measured_pass = m.measure_occlusions && handle_occlusions;
  G<uint8_t> image = as_global(cam.image);
  const uint32_t pitch = cam.pitch;
  const int line = tid;

  // ---- CalculateBasicLineData :1231, IsLineValid :1252, geometric part of CalculateSegmentProbabilities :1441-1455 ----
  bool valid = false, valid_occ = false, horiz = false, reversed = false;
  int start = 0;
  float step = 0.0f, x0 = 0.0f;
  if (line < n_lines) {
    G<v4f> p8 = (G<v4f>)m.points8 + ((uint32_t)view * m.n_points + line) * 2;
    const v4f pa = p8[0], pb4 = p8[1];
    const float cx = pa.x, cy = pa.y, cz = pa.z;
    const float nx = pa.w, ny = pb4.x, nz = pb4.y;
    const float fg = pb4.z, bg = pb4.w;
    float X, Y, Z;
    apply_pose(b2c, cx, cy, cz, X, Y, Z);
    float nu = (b2c.l[0] * nx + b2c.l[3] * ny) + b2c.l[6] * nz;
    float nv = (b2c.l[1] * nx + b2c.l[4] * ny) + b2c.l[7] * nz;
    const float nn = sqrtf(nu * nu + nv * nv);
    if (nn > 0.0f) { nu = nu / nn; nv = nv / nn; }
    const float center_u = X * cam.fu / Z + cam.ppu;
    const float center_v = Y * cam.fv / Z + cam.ppv;
    const float cont = ((fg < bg) ? fg : bg) * cam.fu / (Z * it.fscale);
    valid = !(cont < m.min_continuous_distance) && !(Z <= 0.0f);
    if (valid) {
      const int icu = f2i(center_u + 0.5f), icv = f2i(center_v + 0.5f);
      valid = !(icu < 0 || icu > cam.width - 1 || icv < 0 || icv > cam.height - 1);
    }
    horiz = fabsf(nv) < fabsf(nu);
    float cmaj, cmin, ndom;
    int maj_lim, min_lim1, min_lim2;
    if (horiz) {
      step = nv / nu; cmaj = center_u; cmin = center_v; ndom = nu;
      maj_lim = cam.width - 1; min_lim1 = cam.height - 1; min_lim2 = cam.height - 2;
    } else {
      step = nu / nv; cmaj = center_v; cmin = center_u; ndom = nv;
      maj_lim = cam.height - 1; min_lim1 = cam.width - 1; min_lim2 = cam.width - 2;
    }
    reversed = !(ndom > 0.0f);
    start = f2i(cmaj - it.line_length_half_minus_1);
    const int end = start + it.line_length_minus_1;
    x0 = cmin + step * ((float)start - cmaj) + 0.5f;
    const float xend = x0 + step * (float)it.line_length_minus_1;
    if (valid)
      valid = !(start < 0 || end > maj_lim || f2i(x0) < 0 || f2i(x0) > min_lim1 || f2i(xend) < 1 ||
                f2i(xend) > min_lim2);
    valid_occ = valid;
    if (valid && measured_pass) {  // IsLineUnoccludedMeasured :1343-1389
      float dx, dy, dz;
      apply_pose(b2dc, cx, cy, cz, dx, dy, dz);
      const float du = dx * dcam->fu / dz + dcam->ppu;
      const float dv = dy * dcam->fv / dz + dcam->ppv;
      const float meter_to_pixel = dcam->fu / dz;
      const float diameter = 2.0f * m.measured_occlusion_radius * meter_to_pixel;
      G<float> p = as_global(m.points) + ((size_t)view * m.n_points + line) * M3T_REGION_POINT_FLOATS;
      valid_occ = occlusion_window_clear(*dcam, du, dv, diameter, dz, p[8 + m.measured_depth_offset_id],
                                         m.measured_occlusion_threshold);
    }
    if (valid) {
      state[CS_CX * nl + line] = cx;
      state[CS_CY * nl + line] = cy;
      state[CS_CZ * nl + line] = cz;
      state[CS_CENTER_U * nl + line] = center_u;
      state[CS_CENTER_V * nl + line] = center_v;
      state[CS_NORMAL_U * nl + line] = nu;
      state[CS_NORMAL_V * nl + line] = nv;
      state[CS_NCTS * nl + line] = fabsf(ndom) / it.fscale;
      state[CS_DELTA_R * nl + line] =
          (roundf(cmaj - it.line_length_minus_1_half) + it.line_length_minus_1_half - cmaj) / ndom;
    }
  }
  // two-pass fallback :435-463: the occlusion-handled set only if it has enough lines
  bool take = valid;
  if (measured_pass) {
    const int cnt = wave_sum_i(valid_occ ? 1 : 0);
    int* imisc = reinterpret_cast<int*>(misc);
    if (tid % kWave == 0) imisc[64 + tid / kWave] = cnt;
    __syncthreads();
    int total = 0;
    for (int w = 0; w < (int)blockDim.x / kWave; ++w) total += imisc[64 + w];
    if (total >= m.min_n_unoccluded_lines) take = valid_occ;
  }
  if (line < nl) state[CS_VALID * nl + line] = i2f_bits(take ? 1 : 0);
  if (take) {
    float lf[8], lb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { lf[k] = m.function_lookup_f[k]; lb[k] = m.function_lookup_b[k]; }  // uniform scalars
    float* dist0 = state + CS_DIST0 * nl + line;
    const int bitshift = m.bitshift, bin_bits = 8 - m.bitshift;
    G<v2f> hist = (G<v2f>)m.histogram_norm;
#define M3T_COMPACT_WALK(S, T) \
  case S: compact_walk<S, T>(image, pitch, hist, bitshift, bin_bits, start, step, x0, horiz, reversed, lf, lb, dist0, nl, table); break;
#define M3T_COMPACT_WALKS(T)                                                                                  \
    switch (it.scale) {                                                                                       \
      M3T_COMPACT_WALK(1, T) M3T_COMPACT_WALK(2, T) M3T_COMPACT_WALK(3, T) M3T_COMPACT_WALK(4, T) M3T_COMPACT_WALK(5, T) \
      M3T_COMPACT_WALK(6, T) M3T_COMPACT_WALK(7, T) M3T_COMPACT_WALK(8, T) M3T_COMPACT_WALK(9, T)                 \
      default: break; /* (the host does not choose this kernel for larger scales) */                          \
    }
    if constexpr (TABLE) {
      if (table->fits) {  // (block-uniform: every mixed bin of this object has its slot in LDS)
        M3T_COMPACT_WALKS(true)
      } else {
        M3T_COMPACT_WALKS(false)
      }
    } else {
      M3T_COMPACT_WALKS(false)
    }
#undef M3T_COMPACT_WALKS
#undef M3T_COMPACT_WALK
    // normalisation :1628-1636 and CalculateDistributionMoments :1639-1658 (the thread reads back its own stores)
    float raw[12];
#pragma unroll
    for (int d = 0; d < 12; ++d) raw[d] = dist0[d * nl];
    float area = 0.0f;
#pragma unroll
    for (int d = 0; d < 12; ++d) area += raw[d];
    float dist[12];
#pragma unroll
    for (int d = 0; d < 12; ++d) dist[d] = raw[d] / area;
    float mean_from_begin = 0.0f;
#pragma unroll
    for (int d = 0; d < 12; ++d) mean_from_begin += (float)d * dist[d];
    float var = 0.0f;
#pragma unroll
    for (int d = 0; d < 12; ++d) {
      const float dd = (float)d - mean_from_begin;
      var += (dd * dd) * dist[d];
    }
#pragma unroll
    for (int d = 0; d < 12; ++d) dist0[d * nl] = dist[d];
    state[CS_MEAN * nl + line] = mean_from_begin - m.distribution_length_minus_1_half;
    state[CS_VAR * nl + line] = fmaxf(var, m.min_expected_variance);
  }
  __syncthreads();
  return view;
}

// RegionModality::CalculateGradientAndHessian (:485-558), per line: the eight factors of the line's 27 products.
// rows: [M3T_COMPACT_ROWS][pitch]: J[0..5] | weight * dll | weight / measured_variance | (constants -1)
__device__ __forceinline__ void compact_region_products(CRegion& m, CCam& cam, const Affine& b2c, int corr_iteration,
                                                        int opt_iteration, const float* misc, const float* state,
                                                        int nl, float* rows, int pitch) {
  const RegionIter it = region_iter(m, corr_iteration);
  const int slots = chain_slots(nl);
  for (int line = threadIdx.x; line < slots; line += blockDim.x) {
  float J[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, wg = 0.0f, wh = 0.0f;
  bool ok = false;
  if (line < nl && (f2i_bits(state[CS_VALID * nl + line]) & 1)) {  // :497-513
    ok = true;
    const float cx = state[CS_CX * nl + line], cy = state[CS_CY * nl + line], cz = state[CS_CZ * nl + line];
    float x, y, z;
    apply_pose(b2c, cx, cy, cz, x, y, z);
    const float normal_u = state[CS_NORMAL_U * nl + line], normal_v = state[CS_NORMAL_V * nl + line];
    const float center_u = state[CS_CENTER_U * nl + line], center_v = state[CS_CENTER_V * nl + line];
    const float ncts = state[CS_NCTS * nl + line];
    const float measured_variance = state[CS_VAR * nl + line];
    const float fu_z = cam.fu / z;
    const float fv_z = cam.fv / z;
    const float xfu_z = x * fu_z;
    const float yfv_z = y * fv_z;
    const float delta_cs = (normal_u * (xfu_z + cam.ppu - center_u) + normal_v * (yfv_z + cam.ppv - center_v) -
                            state[CS_DELTA_R * nl + line]) *
                           ncts;
    float dll = 0.0f;
    if (opt_iteration < m.n_global_iterations) {
      dll = (state[CS_MEAN * nl + line] - delta_cs) / measured_variance;
    } else {
      const int upper = f2i(delta_cs + m.distribution_length_plus_1_half);
      const int lower = upper - 1;
      if (upper <= 0 || upper >= 12) {
        ok = false;
      } else {
        const float d_upper = state[(CS_DIST0 + upper) * nl + line], d_lower = state[(CS_DIST0 + lower) * nl + line];
        typedef const __attribute__((address_space(3))) double* LdsDoubles;
        LdsDoubles log_table = (LdsDoubles)(misc + kMiscLogTable);
        float log_upper = 0.0f, log_lower = 0.0f;
        const bool vouched = m3t_log_fast(d_upper, log_table, &log_upper) & m3t_log_fast(d_lower, log_table, &log_lower);
        if (!vouched) {
          log_upper = (float)log((double)d_upper);
          log_lower = (float)log((double)d_lower);
        }
        dll = (log_upper - log_lower) * m.learning_rate / measured_variance;
      }
    }
    if (ok) {
      const float dc0 = ncts * normal_u * fu_z;
      const float dc1 = ncts * normal_v * fv_z;
      const float dc2 = ncts * (-normal_u * xfu_z - normal_v * yfv_z) / z;
      const float t0 = (dc0 * b2c.l[0] + dc1 * b2c.l[1]) + dc2 * b2c.l[2];
      const float t1 = (dc0 * b2c.l[3] + dc1 * b2c.l[4]) + dc2 * b2c.l[5];
      const float t2 = (dc0 * b2c.l[6] + dc1 * b2c.l[7]) + dc2 * b2c.l[8];
      J[0] = cy * t2 - cz * t1;
      J[1] = cz * t0 - cx * t2;
      J[2] = cx * t1 - cy * t0;
      J[3] = t0;
      J[4] = t1;
      J[5] = t2;
      const float weight = m.min_expected_variance / (ncts * ncts * it.variance);
      wg = weight * dll;
      wh = weight / measured_variance;
    }
  }
  float* out = rows + line;
  // a slot that does not contribute stores zeros: (0 * 0) * c = +-0, and x - (+-0) = x for the running sums
#pragma unroll
  for (int r = 0; r < 6; ++r) out[r * pitch] = ok ? J[r] : 0.0f;
  out[6 * pitch] = ok ? wg : 0.0f;
  out[7 * pitch] = ok ? wh : 0.0f;
  }
}

// DepthModality::CalculateGradientAndHessian (depth_modality.cpp:333-381), per point: v = [p x n; n], weight, weight^2 eps
__device__ __forceinline__ void compact_depth_products(CDepth& m, const Affine& b2c, int corr_iteration, const float* ps,
                                                       int np, float* rows, int pitch) {
  const int slots = chain_slots(np);
  for (int i = threadIdx.x; i < slots; i += blockDim.x) {
  float v[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, weight = 0.0f, se = 0.0f;
  if (i < np && (f2i_bits(ps[PS_VALID * np + i]) & 1)) {
    const Affine c2b = inverse_pose(b2c);
    const float standard_deviation = last_valid(m.standard_deviations, m.n_standard_deviations, corr_iteration);
    float qx, qy, qz;
    apply_pose(c2b, ps[PS_CORR_X * np + i], ps[PS_CORR_Y * np + i], ps[PS_CORR_Z * np + i], qx, qy, qz);
    const float nx = ps[PS_NX * np + i], ny = ps[PS_NY * np + i], nz = ps[PS_NZ * np + i];
    const float d0 = ps[PS_CX * np + i] - qx, d1 = ps[PS_CY * np + i] - qy, d2 = ps[PS_CZ * np + i] - qz;
    const float epsilon = (nx * d0 + ny * d1) + nz * d2;
    v[0] = qy * nz - qz * ny;
    v[1] = qz * nx - qx * nz;
    v[2] = qx * ny - qy * nx;
    v[3] = nx;
    v[4] = ny;
    v[5] = nz;
    weight = 1.0f / (standard_deviation * ps[PS_CORR_Z * np + i]);
    se = (weight * weight) * epsilon;
  }
  float* out = rows + i;
#pragma unroll
  for (int r = 0; r < 6; ++r) out[r * pitch] = v[r];
  out[6 * pitch] = weight;
  out[7 * pitch] = se;
  }
}

// The 42 running sums (layout of Modality::gradient() / hessian()) in the reference's line-after-line order,
// region_modality.cpp:550-554 / depth_modality.cpp:361-377, from the factor rows:
//   region   gradient r:  s -= (wg * J[r]) * (-1)        == s -= -(wg * J[r])           (gradient_ += (weight dll) J^T)
//            Hessian r>=c: s -= (wh * J[r]) * J[c]                                        (hessian_ -= ((w / var) J^T) J)
//   depth    gradient r:  s -= (se * v[r]) * (1 * 1)                                      (gradient_ -= (w^2 eps) v)
//            Hessian c<=r: s -= (weight * v[c]) * (weight * v[r])                          (hessian_ -= (w v)(w v)^T)
// Lanes >= 42 run along with lane 0's rows and are ignored.
__device__ __forceinline__ void compact_chain(const float* rows_r, int pitch_r, int slots_r, const float* rows_d,
                                              int pitch_d, int slots_d, int lane, float& sum_r, float& sum_d) {
  int ia, ib, ic, da, db, dc, dd;
  if (lane < 6 || lane >= 42) {
    const int l = lane < 6 ? lane : 0;
    ia = 6; ib = l; ic = 8;
    da = 7; db = l; dc = 8; dd = 8;
  } else {
    const int idx = lane - 6, c = idx / 6, r = idx - c * 6;
    const int lo = r >= c ? r : c, hi = r >= c ? c : r;  // lo: the larger index
    ia = 7; ib = lo; ic = hi;
    da = 6; db = hi; dc = 6; dd = lo;
  }
  float sr = 0.0f, sd = 0.0f;
  if (rows_r) {
    LdsV4 pa = (LdsV4)(rows_r + ia * pitch_r), pb = (LdsV4)(rows_r + ib * pitch_r), pc = (LdsV4)(rows_r + ic * pitch_r);
    const int nq = slots_r / 4;  // a multiple of 6
    for (int q = 0; q < nq; q += 2) {
      const v4f a0 = pa[q], b0 = pb[q], c0 = pc[q];
      const v4f a1 = pa[q + 1], b1 = pb[q + 1], c1 = pc[q + 1];
      const v4f p0 = (a0 * b0) * c0, p1 = (a1 * b1) * c1;
      sr -= p0.x; sr -= p0.y; sr -= p0.z; sr -= p0.w;
      sr -= p1.x; sr -= p1.y; sr -= p1.z; sr -= p1.w;
    }
  }
  if (rows_d) {
    LdsV4 pa = (LdsV4)(rows_d + da * pitch_d), pb = (LdsV4)(rows_d + db * pitch_d), pc = (LdsV4)(rows_d + dc * pitch_d),
          pd = (LdsV4)(rows_d + dd * pitch_d);
    const int nq = slots_d / 4;
    for (int q = 0; q < nq; q += 2) {
      const v4f a0 = pa[q], b0 = pb[q], c0 = pc[q], d0 = pd[q];
      const v4f a1 = pa[q + 1], b1 = pb[q + 1], c1 = pc[q + 1], d1 = pd[q + 1];
      const v4f p0 = (a0 * b0) * (c0 * d0), p1 = (a1 * b1) * (c1 * d1);
      sd -= p0.x; sd -= p0.y; sd -= p0.z; sd -= p0.w;
      sd -= p1.x; sd -= p1.y; sd -= p1.z; sd -= p1.w;
    }
  }
  sum_r = sr;
  sum_d = sd;
}

}  // namespace

extern "C" {

// One 256-thread workgroup per rigid optimizer, four or five per CU.
#ifndef M3T_COMPACT_WAVES
#define M3T_COMPACT_WAVES 4  /* waves per SIMD the register budget is held to (4 x 256-thread workgroups per CU) */
#endif
__global__ void __launch_bounds__(M3T_COMPACT_THREADS, M3T_COMPACT_WAVES)
tracking_step_compact_kernel(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                             const CameraDev* cams, float* body_poses, CompactLayout L, int iteration,
                             int n_corr_iterations, int n_update_iterations, int fuse_histogram) {
  const RoiGuardArgs guard{};  // (unused)
#define M3T_COMPACT_GUARD false
#define M3T_COMPACT_TABLE false
#include "m3t_compact_step.inc"
#undef M3T_COMPACT_TABLE
#undef M3T_COMPACT_GUARD
}
// ... with 512-thread workgroups, two per CU (the same registers and LDS per object): batches of at most two objects per CU
// whose step is the depth scan -- sixteen lanes per point, 200 points: 12.5 rounds of a 256-thread workgroup, half of that here
__global__ void __launch_bounds__(2 * M3T_COMPACT_THREADS, M3T_COMPACT_WAVES)
tracking_step_compact_wide_kernel(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                                  const CameraDev* cams, float* body_poses, CompactLayout L, int iteration,
                                  int n_corr_iterations, int n_update_iterations, int fuse_histogram) {
  const RoiGuardArgs guard{};  // (unused)
#define M3T_COMPACT_GUARD false
#define M3T_COMPACT_TABLE false
#include "m3t_compact_step.inc"
#undef M3T_COMPACT_TABLE
#undef M3T_COMPACT_GUARD
}
// ... reading frame slots that hold the trackers' rectangles only (ROI ingest)
__global__ void __launch_bounds__(M3T_COMPACT_THREADS, M3T_COMPACT_WAVES)
tracking_step_compact_guard_kernel(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                                   const CameraDev* cams, float* body_poses, CompactLayout L, int iteration,
                                   int n_corr_iterations, int n_update_iterations, int fuse_histogram, RoiGuardArgs guard) {
#define M3T_COMPACT_GUARD true
#define M3T_COMPACT_TABLE false
#include "m3t_compact_step.inc"
#undef M3T_COMPACT_TABLE
#undef M3T_COMPACT_GUARD
}

// ... with the pair table compacted in LDS (three workgroups per CU: the table costs the fourth one its room)
__global__ void __launch_bounds__(M3T_COMPACT_THREADS, M3T_TABLE_WAVES)
tracking_step_compact_table_kernel(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                                   const CameraDev* cams, float* body_poses, CompactLayout L, int iteration,
                                   int n_corr_iterations, int n_update_iterations, int fuse_histogram) {
  const RoiGuardArgs guard{};  // (unused)
#define M3T_COMPACT_GUARD false
#define M3T_COMPACT_TABLE true
#include "m3t_compact_step.inc"
#undef M3T_COMPACT_TABLE
#undef M3T_COMPACT_GUARD
}

}  // extern "C"
