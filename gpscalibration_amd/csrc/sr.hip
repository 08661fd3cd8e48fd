// sr.hip -- scanRegistration's feature extraction and pcl::VoxelGrid for many sweeps /
// clouds at once.  One workgroup (512 threads) owns one sweep: ring / time tagging, the
// stable split into 16 rings, curvature, occlusion rejection, and then the 96 (ring,
// sector) picking rounds, which are order dependent (marks spill +-5 points across sector
// borders) and therefore run one after the other inside the workgroup: a bitonic sort of
// the sector in LDS, then a wave-cooperative greedy pick (64 candidates per ballot).
// All per-point flags live in LDS bitsets.  gfx950 only.
//
// Replaces, in /root/reference/src/gpsCalibration/src/lidar_slam/loam/scanRegistration.cpp:
//   removeNaN, start/end orientation       :262-280
//   ring id, orientation, relative time    :284-363 (IMU block :364-433 inactive under run.sh)
//   ring concatenation                     :444-447
//   curvature + ring spans                 :455-490
//   occluded / parallel-beam rejection     :492-548
//   per-sector sort and picking            :558-665
//   VoxelGrid 0.2 of the less-flat points  :667-673
// and pcl::VoxelGrid<PointXYZI>::filter as used at SR:667-673 and laserMapping.cpp:1044-1058.
// The reference's common.h says `using namespace std;`, so its unqualified sqrt / atan / atan2
// on floats are the float32 overloads; atanf / atan2f are evaluated here as the correctly
// rounded float32 of the float64 function; no contraction anywhere.
#include "block_utils.hpp"
#include "common.hpp"
#include "loam_internal.hpp"
#include "wave_reduce.hpp"

#include <algorithm>

namespace gpscal {

struct SrDesc {
    long long in_off;  // first input point; full / scratch arrays use the same offset
    long long lf_off;  // first less-flat output slot
    long long key_off; // first slot of the global sort scratch
    int n_in, lf_cap, key_cap, pad;
};

struct VgDesc {
    long long off, key_off;
    int n, key_cap;
};

__global__ __launch_bounds__(SBLOCK) void voxel_grid_kernel(const VgDesc *__restrict__ descs,
                                                            const float4 *__restrict__ src, float leaf,
                                                            float4 *__restrict__ dst, unsigned long long *g_keys,
                                                            int *__restrict__ counts, int *__restrict__ status)
{
    extern __shared__ unsigned long long dyn_lds[];
    __shared__ BlockShared S;
    __shared__ int s_count;
    const VgDesc D = descs[blockIdx.x];
    if (threadIdx.x == 0) {
        s_count = 0;
        S.overflow = 0;
    }
    __syncthreads();
    block_voxel_grid(S, src + D.off, D.n, leaf, dst + D.off, D.n, &s_count, dyn_lds, g_keys + D.key_off, D.key_cap);
    __syncthreads();
    if (threadIdx.x == 0) {
        counts[blockIdx.x] = s_count;
        if (S.overflow) atomicOr(status, 1);
    }
}

// ---------------------------------------------------------------- scanRegistration
// marks the +-5 neighbourhood of a picked point (SR:597-616, 634-655); wave 0, all lanes
__device__ __forceinline__ void mark_neighbours(unsigned *picked, unsigned *gap, int ind, int cloud_size, int lane)
{
    // lanes 0..4: l = 1..5 forward (break on gap[ind+l-1]); lanes 8..12: l = -1..-5 backward (gap[ind+l])
    bool brk = false;
    int tgt = -1;
    if (lane < 5) {
        const int l = lane + 1;
        tgt = ind + l;
        brk = tgt >= cloud_size || get_bit(gap, tgt - 1);
    } else if (lane >= 8 && lane < 13) {
        const int l = -(lane - 7);
        tgt = ind + l;
        brk = tgt < 0 || get_bit(gap, tgt);
    }
    const unsigned long long m = __ballot(brk);
    const unsigned fwd = (unsigned)(m & 0x1f), bwd = (unsigned)((m >> 8) & 0x1f);
    const int nf = fwd ? __ffs(fwd) - 1 : 5, nb = bwd ? __ffs(bwd) - 1 : 5;
    if (lane == 0) set_bit(picked, ind);
    if (lane < 5 && lane < nf) set_bit(picked, tgt);
    if (lane >= 8 && lane < 13 && lane - 8 < nb) set_bit(picked, tgt);
    __builtin_amdgcn_wave_barrier();
}

// One wave's greedy pick of one sorted sector (SR:578-657): K[0, cnt) = (curvature bits, point index), ascending.
// cnt3 = {sharp, less sharp, flat} points written so far to o_sharp / o_lsharp / o_flat (LDS counters).
__device__ __forceinline__ void sr_pick_sector(const unsigned long long *K, int cnt, unsigned *picked, unsigned *gap,
                                               unsigned *labpos, const float4 *__restrict__ cloud,
                                               float4 *__restrict__ o_sharp, float4 *__restrict__ o_lsharp,
                                               float4 *__restrict__ o_flat, int *cnt3, int cs, int lane)
{
    int n_sharp = cnt3[0], n_lsharp = cnt3[1], n_flat = cnt3[2];
    // corners: from the largest curvature down (SR:578-619)
    int largest = 0;
    bool done = false;
    for (int top = cnt - 1; top >= 0 && !done; top -= 64) {
        const int k = top - lane;
        const bool valid = k >= 0;
        const unsigned long long key = valid ? K[k] : 0ull;
        const int ind = (int)(unsigned)key;
        const bool cand = valid && (double)__uint_as_float((unsigned)(key >> 32)) > 0.1;
        while (true) {
            const unsigned long long m = __ballot(cand && !get_bit(picked, ind));
            if (!m) break;
            const int f = __ffsll((long long)m) - 1;
            const int indf = __shfl(ind, f);
            ++largest;
            if (largest > 20) {
                done = true;
                break;
            }
            if (lane == 0) {
                const float4 pt = cloud[indf];
                if (largest <= 16) o_sharp[n_sharp] = pt;
                o_lsharp[n_lsharp] = pt;
                set_bit(labpos, indf);
            }
            if (largest <= 16) ++n_sharp;
            ++n_lsharp;
            mark_neighbours(picked, gap, indf, cs, lane);
        }
    }
    // flat points: from the smallest curvature up (SR:621-657)
    int smallest = 0;
    done = false;
    for (int bot = 0; bot < cnt && !done; bot += 64) {
        const int k = bot + lane;
        const bool valid = k < cnt;
        const unsigned long long key = valid ? K[k] : 0ull;
        const int ind = (int)(unsigned)key;
        const bool cand = valid && (double)__uint_as_float((unsigned)(key >> 32)) < 0.1;
        while (true) {
            const unsigned long long m = __ballot(cand && !get_bit(picked, ind));
            if (!m) break;
            const int f = __ffsll((long long)m) - 1;
            const int indf = __shfl(ind, f);
            if (lane == 0) o_flat[n_flat] = cloud[indf];
            ++n_flat;
            ++smallest;
            if (smallest >= 32) {
                done = true;
                break;
            }
            mark_neighbours(picked, gap, indf, cs, lane);
        }
    }
    if (lane == 0) {
        cnt3[0] = n_sharp;
        cnt3[1] = n_lsharp;
        cnt3[2] = n_flat;
    }

}

// Stable ascending sort by curvature of one sector (the insertion sort of SR:567-575) by ONE wave, in that wave's own
// key buffers: sid[sp, sp + cnt) is rewritten in sorted order and K[k] = (curvature bits, point index).
__device__ __forceinline__ void sr_wave_sort_sector(unsigned long long *K, int *OLD, const float *__restrict__ cv,
                                                    int *__restrict__ sid, int sp, int cnt, int lane)
{
    const int np2 = next_pow2(cnt);
    for (int k = lane; k < np2; k += 64) {
        unsigned long long key = ~0ull;
        if (k < cnt) {
            const int ind = sid[sp + k];
            OLD[k] = ind;
            key = ((unsigned long long)__float_as_uint(cv[ind]) << 32) | (unsigned)k;
        }
        K[k] = key;
    }
    __builtin_amdgcn_wave_barrier();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (np2 >> 1); t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const bool up = (i & k) == 0;
                const unsigned long long a = K[i], b = K[p];
                if ((a > b) == up) {
                    K[i] = b;
                    K[p] = a;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    for (int k = lane; k < cnt; k += 64) {
        const unsigned long long key = K[k];
        const int ind = OLD[(unsigned)key];
        K[k] = (key & 0xffffffff00000000ull) | (unsigned)ind;
        sid[sp + k] = ind;
    }
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(SBLOCK) void scan_registration_kernel(
    const SrDesc *__restrict__ descs, const float *__restrict__ xyz, float4 *__restrict__ full,
    float4 *__restrict__ sharp, float4 *__restrict__ less_sharp, float4 *__restrict__ flat,
    float4 *__restrict__ less_flat, signed char *__restrict__ ringbuf, float *__restrict__ oribuf,
    float *__restrict__ curv, int *__restrict__ sort_ind, float4 *__restrict__ lfs, unsigned long long *g_keys,
    int *g_old, int *__restrict__ counts, int *__restrict__ status, int *__restrict__ ring_counts)
{
    extern __shared__ unsigned long long dyn_lds[];
    unsigned long long *lds_keys = dyn_lds;                              // LDS_KEYS
    int *lds_old = reinterpret_cast<int *>(dyn_lds + LDS_KEYS);          // LDS_KEYS
    unsigned *picked = reinterpret_cast<unsigned *>(lds_old + LDS_KEYS); // SR_BIT_WORDS each
    unsigned *gap = picked + SR_BIT_WORDS;
    unsigned *labpos = gap + SR_BIT_WORDS;
    __shared__ BlockShared S;
    __shared__ int s_nl;

    const int b = blockIdx.x;
    const SrDesc D = descs[b];
    const float *in = xyz + 3 * D.in_off;
    float4 *cloud = full + D.in_off;
    signed char *rings = ringbuf + D.in_off;
    float *oris = oribuf + D.in_off;
    float *cv = curv + D.in_off;
    int *sid = sort_ind + D.in_off;
    float4 *lf = lfs + D.in_off;
    float4 *o_sharp = sharp + (long long)b * 1536, *o_lsharp = less_sharp + (long long)b * 1920,
           *o_flat = flat + (long long)b * 3072, *o_lflat = less_flat + D.lf_off;
    unsigned long long *gk = g_keys + D.key_off;
    int *go = g_old + D.key_off;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = D.n_in;

    if (threadIdx.x == 0) {
        S.first_fin = 0x7fffffff;
        S.last_fin = -1;
        S.hp_idx = 0x7fffffff;
        S.overflow = 0;
        for (int k = 0; k < 5; ++k) S.counts[k] = 0;
    }
    if (threadIdx.x < N_RINGS) {
        S.ring_cnt[threadIdx.x] = 0;
        S.bnd[threadIdx.x] = -1;
    }
    for (int i = threadIdx.x; i < 3 * SR_BIT_WORDS; i += SBLOCK) picked[i] = 0u;
    __syncthreads();

    // ---- first / last finite point (removeNaNFromPointCloud, SR:265-266)
    {
        int f = 0x7fffffff, l = -1;
        for (int i = threadIdx.x; i < n; i += SBLOCK)
            if (fin3(in[3 * i], in[3 * i + 1], in[3 * i + 2])) {
                f = min(f, i);
                l = max(l, i);
            }
        if (f != 0x7fffffff) {
            atomicMin(&S.first_fin, f);
            atomicMax(&S.last_fin, l);
        }
    }
    __syncthreads();
    if (S.last_fin < 0) {  // nothing but NaNs
        if (threadIdx.x < 5) counts[5 * b + threadIdx.x] = 0;
        return;
    }
    if (threadIdx.x == 0) {
        const float *p0 = in + 3 * S.first_fin, *p1 = in + 3 * S.last_fin;
        const float so = -atan2_f(p0[1], p0[0]);                               // SR:270
        float eo = (float)((double)(-atan2_f(p1[1], p1[0])) + 2 * PI_D);       // SR:272-273
        if ((double)(eo - so) > 3 * PI_D) eo = (float)((double)eo - 2 * PI_D); // SR:277-281
        else if ((double)(eo - so) < PI_D) eo = (float)((double)eo + 2 * PI_D);
        S.start_ori = so;
        S.end_ori = eo;
    }
    __syncthreads();
    const float startOri = S.start_ori, endOri = S.end_ori;

    // ---- ring id and orientation per point; where halfPassed flips (SR:284-351)
    {
        int hp = 0x7fffffff;
        for (int i = threadIdx.x; i < n; i += SBLOCK) {
            const float ix = in[3 * i], iy = in[3 * i + 1], iz = in[3 * i + 2];
            int r = -1;
            float ori = 0.f;
            if (fin3(ix, iy, iz)) {
                const float px = iy, py = iz, pz = ix;  // SR:295-297
                const float angle = (float)((double)(atan_f(py / sqrtf(px * px + pz * pz)) * 180) / PI_D);
                const int rounded = (int)((double)angle + (angle < 0.0f ? -0.5 : +0.5));
                r = ring_of(rounded);
                if (r >= 0) {
                    ori = -atan2_f(px, pz);
                    float o1 = ori;
                    if ((double)o1 < (double)startOri - PI_D / 2) o1 = (float)((double)o1 + 2 * PI_D);
                    else if ((double)o1 > (double)startOri + PI_D * 3 / 2) o1 = (float)((double)o1 - 2 * PI_D);
                    if ((double)(o1 - startOri) > PI_D) hp = min(hp, i);
                    atomicAdd(&S.ring_cnt[r], 1);
                }
            }
            rings[i] = (signed char)r;
            oris[i] = ori;
        }
        if (hp != 0x7fffffff) atomicMin(&S.hp_idx, hp);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int r = 0; r < N_RINGS; ++r) {
            S.base[r] = acc;
            acc += S.ring_cnt[r];
        }
        S.base[N_RINGS] = acc;
        S.cloud_size = acc;
    }
    __syncthreads();
    const int cs = S.cloud_size;
    if (cs > SR_MAX_POINTS) {  // the reference's work arrays end here (common.h:15)
        if (threadIdx.x == 0) atomicOr(status, 2);
        if (threadIdx.x < 5) counts[5 * b + threadIdx.x] = 0;
        return;
    }
    // ---- stable split into rings + time tag (SR:352-363, 444-447)
    {
        const int hp_idx = S.hp_idx;
        for (int i0 = 0; i0 < n; i0 += SBLOCK) {
            const int i = i0 + threadIdx.x;
            int r = -1;
            if (i < n) r = rings[i];
            int myrank = 0;
#pragma unroll
            for (int rr = 0; rr < N_RINGS; ++rr) {
                const unsigned long long m = __ballot(r == rr);
                if (r == rr) myrank = __popcll(m & ((1ull << lane) - 1ull));
                if (lane == 0) S.wcnt[wave][rr] = __popcll(m);
            }
            __syncthreads();
            if (r >= 0) {
                int pos = S.base[r] + myrank;
                for (int w = 0; w < wave; ++w) pos += S.wcnt[w][r];
                float ori = oris[i];
                if (i <= hp_idx) {  // the point that flips halfPassed is itself still "before"
                    if ((double)ori < (double)startOri - PI_D / 2) ori = (float)((double)ori + 2 * PI_D);
                    else if ((double)ori > (double)startOri + PI_D * 3 / 2) ori = (float)((double)ori - 2 * PI_D);
                } else {
                    ori = (float)((double)ori + 2 * PI_D);
                    if ((double)ori < (double)endOri - PI_D * 3 / 2) ori = (float)((double)ori + 2 * PI_D);
                    else if ((double)ori > (double)endOri + PI_D / 2) ori = (float)((double)ori - 2 * PI_D);
                }
                const float relTime = (ori - startOri) / (endOri - startOri);
                cloud[pos] = make_float4(in[3 * i + 1], in[3 * i + 2], in[3 * i], (float)(r + 0.1 * (double)relTime));
            }
            __syncthreads();
            if (threadIdx.x < N_RINGS) {
                int add = 0;
                for (int w = 0; w < SWAVES; ++w) add += S.wcnt[w][threadIdx.x];
                S.base[threadIdx.x] += add;
            }
            __syncthreads();
        }
    }
    // ---- curvature, ring spans, gaps, occlusion rejection (SR:455-548)
    for (int i0 = 0; i0 < cs; i0 += SBLOCK) {
        const int i = i0 + threadIdx.x;
        bool g = false;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f), pn = p;
        float diff = 0.f;
        if (i < cs) p = cloud[i];
        if (i + 1 < cs) {
            pn = cloud[i + 1];
            const float dX = pn.x - p.x, dY = pn.y - p.y, dZ = pn.z - p.z;
            diff = dX * dX + dY * dY + dZ * dZ;
            g = (double)diff > 0.05;
        }
        const unsigned long long gm = __ballot(g);
        if (lane == 0 && i0 + 64 * wave < cs) {
            gap[(i0 >> 5) + 2 * wave] = (unsigned)gm;
            gap[(i0 >> 5) + 2 * wave + 1] = (unsigned)(gm >> 32);
        }
        int sidv = 0;
        float c = 0.f;
        if (i >= 5 && i < cs - 5) {
            float4 q[11];
#pragma unroll
            for (int k = 0; k < 11; ++k) q[k] = cloud[i - 5 + k];
            const float dx = q[0].x + q[1].x + q[2].x + q[3].x + q[4].x - 10 * q[5].x + q[6].x + q[7].x + q[8].x + q[9].x + q[10].x;
            const float dy = q[0].y + q[1].y + q[2].y + q[3].y + q[4].y - 10 * q[5].y + q[6].y + q[7].y + q[8].y + q[9].y + q[10].y;
            const float dz = q[0].z + q[1].z + q[2].z + q[3].z + q[4].z - 10 * q[5].z + q[6].z + q[7].z + q[8].z + q[9].z + q[10].z;
            c = dx * dx + dy * dy + dz * dz;
            sidv = i;
            const int ci = (int)p.w;
            const int cprev = i == 5 ? -1 : (int)q[4].w;
            if (ci != cprev && ci > 0 && ci < N_RINGS) atomicMax(&S.bnd[ci], i);  // last change wins (SR:480-487)
        }
        if (i < cs) {
            cv[i] = c;
            sid[i] = sidv;
        }
        if (i >= 5 && i < cs - 6) {  // SR:492-548
            if ((double)diff > 0.1) {
                const float depth1 = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
                const float depth2 = sqrtf(pn.x * pn.x + pn.y * pn.y + pn.z * pn.z);
                if (depth1 > depth2) {
                    const float dX = pn.x - p.x * depth2 / depth1, dY = pn.y - p.y * depth2 / depth1,
                                dZ = pn.z - p.z * depth2 / depth1;
                    if ((double)(sqrtf(dX * dX + dY * dY + dZ * dZ) / depth2) < 0.1)
                        for (int l = -5; l <= 0; ++l) set_bit(picked, i + l);
                } else {
                    const float dX = pn.x * depth1 / depth2 - p.x, dY = pn.y * depth1 / depth2 - p.y,
                                dZ = pn.z * depth1 / depth2 - p.z;
                    if ((double)(sqrtf(dX * dX + dY * dY + dZ * dZ) / depth1) < 0.1)
                        for (int l = 1; l <= 6; ++l) set_bit(picked, i + l);
                }
            }
            const float4 pp = cloud[i - 1];
            const float eX = p.x - pp.x, eY = p.y - pp.y, eZ = p.z - pp.z;
            const float diff2 = eX * eX + eY * eY + eZ * eZ;
            const float dis = p.x * p.x + p.y * p.y + p.z * p.z;
            if ((double)diff > 0.0002 * (double)dis && (double)diff2 > 0.0002 * (double)dis) set_bit(picked, i);
        }
    }
    __syncthreads();
    if (threadIdx.x < N_RINGS) {
        S.scan_start[threadIdx.x] = 0;
        S.scan_end[threadIdx.x] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int c = 1; c < N_RINGS; ++c)
            if (S.bnd[c] >= 0) {
                S.scan_start[c] = S.bnd[c] + 5;
                S.scan_end[c - 1] = S.bnd[c] - 5;
            }
        S.scan_start[0] = 5;
        S.scan_end[N_RINGS - 1] = cs - 5;
    }
    __syncthreads();

    // ---- picking (SR:558-674).  The 96 (ring, sector) rounds are order dependent only inside a ring (marks spill
    // +-5 points across sector borders, never across a ring's own +-5 margin), and a round is one wave's work (~50
    // sequential ballots): when every sector fits a wave's share of the LDS key buffer, the eight waves pick two
    // rings each side by side -- into the ring's own stretch of the output arrays, which are sized for exactly that
    // (16 x 96 sharp, 16 x 120 less sharp, 16 x 192 flat) -- and the lists are closed up in ring order afterwards.
    __shared__ int s_par;
    __shared__ int r_cnt[N_RINGS][3];
    __shared__ int r_pre[3][N_RINGS + 1];
    if (threadIdx.x == 0) {
        int ok = 1;
        for (int ring = 0; ring < N_RINGS; ++ring)
            for (int j = 0; j < 6; ++j) {
                const int sp = (S.scan_start[ring] * (6 - j) + S.scan_end[ring] * j) / 6;
                const int ep = (S.scan_start[ring] * (5 - j) + S.scan_end[ring] * (j + 1)) / 6 - 1;
                if (ep - sp + 1 > LDS_KEYS / SWAVES) ok = 0;
            }
        s_par = ok;
    }
    if (threadIdx.x < N_RINGS * 3) r_cnt[threadIdx.x / 3][threadIdx.x % 3] = 0;
    __syncthreads();
    if (s_par) {
        unsigned long long *Kw = lds_keys + wave * (LDS_KEYS / SWAVES);
        int *Ow = lds_old + wave * (LDS_KEYS / SWAVES);
        for (int ring = wave; ring < N_RINGS; ring += SWAVES)
            for (int j = 0; j < 6; ++j) {
                const int sp = (S.scan_start[ring] * (6 - j) + S.scan_end[ring] * j) / 6;
                const int ep = (S.scan_start[ring] * (5 - j) + S.scan_end[ring] * (j + 1)) / 6 - 1;
                const int cnt = ep - sp + 1;
                if (cnt <= 0) continue;
                sr_wave_sort_sector(Kw, Ow, cv, sid, sp, cnt, lane);
                sr_pick_sector(Kw, cnt, picked, gap, labpos, cloud, o_sharp + ring * 96, o_lsharp + ring * 120,
                               o_flat + ring * 192, r_cnt[ring], cs, lane);
            }
        __syncthreads();
        if (threadIdx.x < 3) {
            int acc = 0;
            for (int r = 0; r < N_RINGS; ++r) {
                r_pre[threadIdx.x][r] = acc;
                acc += r_cnt[r][threadIdx.x];
            }
            r_pre[threadIdx.x][N_RINGS] = acc;
        }
        __syncthreads();
        {  // close the three lists up: every element is read before any is written
            float4 v[3][6];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float4 *base = t == 0 ? o_sharp : (t == 1 ? o_lsharp : o_flat);
                const int cap = t == 0 ? 96 : (t == 1 ? 120 : 192);
                const int tot = r_pre[t][N_RINGS];
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int e = threadIdx.x + u * SBLOCK;
                    v[t][u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e < tot) {
                        int r = 0;
                        while (r + 1 < N_RINGS && r_pre[t][r + 1] <= e) ++r;
                        v[t][u] = base[r * cap + (e - r_pre[t][r])];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float4 *base = t == 0 ? o_sharp : (t == 1 ? o_lsharp : o_flat);
                const int tot = r_pre[t][N_RINGS];
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int e = threadIdx.x + u * SBLOCK;
                    if (e < tot) base[e] = v[t][u];
                }
            }
        }
        if (threadIdx.x == 0) {
            S.counts[1] = r_pre[0][N_RINGS];
            S.counts[2] = r_pre[1][N_RINGS];
            S.counts[3] = r_pre[2][N_RINGS];
        }
        __syncthreads();
        // the less-flat points ring by ring: everything that is not a corner (SR:659-663), then VoxelGrid 0.2
        for (int ring = 0; ring < N_RINGS; ++ring) {
            const int sp = S.scan_start[ring], ep = S.scan_end[ring] - 1;  // the six sectors are contiguous
            int nl = 0;
            const int before = S.counts[4];
            __syncthreads();
            for (int k0 = sp; k0 <= ep; k0 += SBLOCK) {
                const int k = k0 + threadIdx.x;
                const bool keep = k <= ep && !get_bit(labpos, k);
                int tot;
                const int r = block_rank(S, keep, tot);
                if (keep) lf[nl + r] = cloud[k];
                nl += tot;
            }
            __syncthreads();
            block_voxel_grid(S, lf, nl, 0.2f, o_lflat, D.lf_cap, &S.counts[4], lds_keys, gk, D.key_cap);  // SR:667-673
            __syncthreads();
            if (threadIdx.x == 0 && ring_counts) {
                ring_counts[32 * b + ring] = r_cnt[ring][1];
                ring_counts[32 * b + 16 + ring] = S.counts[4] - before;
            }
        }
    } else {
        // ---- picking, ring by ring and sector by sector (SR:558-674)
        for (int ring = 0; ring < N_RINGS; ++ring) {
            if (threadIdx.x == 0) {
                s_nl = 0;
                if (ring_counts) {  // less-sharp / less-flat points emitted by this ring index so far
                    ring_counts[32 * b + ring] = -S.counts[2];
                    ring_counts[32 * b + 16 + ring] = -S.counts[4];
                }
            }
            __syncthreads();
            for (int j = 0; j < 6; ++j) {
                const int sp = (S.scan_start[ring] * (6 - j) + S.scan_end[ring] * j) / 6;
                const int ep = (S.scan_start[ring] * (5 - j) + S.scan_end[ring] * (j + 1)) / 6 - 1;
                const int cnt = ep - sp + 1;
                if (cnt <= 0) continue;  // uniform
                const int np2 = next_pow2(cnt);
                const bool in_lds = np2 <= LDS_KEYS;
                unsigned long long *K = in_lds ? lds_keys : gk;
                int *OLD = in_lds ? lds_old : go;
                if (!in_lds && np2 > D.key_cap) {
                    if (threadIdx.x == 0) S.overflow = 1;
                    continue;
                }
                // stable ascending sort by curvature = the insertion sort of SR:567-575
                for (int k = threadIdx.x; k < np2; k += SBLOCK) {
                    unsigned long long key = ~0ull;
                    if (k < cnt) {
                        const int ind = sid[sp + k];
                        OLD[k] = ind;
                        key = ((unsigned long long)__float_as_uint(cv[ind]) << 32) | (unsigned)k;
                    }
                    K[k] = key;
                }
                __syncthreads();
                block_bitonic_sort(K, np2);
                for (int k = threadIdx.x; k < cnt; k += SBLOCK) {
                    const unsigned long long key = K[k];
                    const int ind = OLD[(unsigned)key];
                    K[k] = (key & 0xffffffff00000000ull) | (unsigned)ind;
                    sid[sp + k] = ind;
                }
                __syncthreads();
                if (wave == 0) sr_pick_sector(K, cnt, picked, gap, labpos, cloud, o_sharp, o_lsharp, o_flat, &S.counts[1], cs, lane);
                __syncthreads();
                // everything of the sector that is not a corner (SR:659-663)
                int nl = s_nl;
                __syncthreads();
                for (int k0 = sp; k0 <= ep; k0 += SBLOCK) {
                    const int k = k0 + threadIdx.x;
                    const bool keep = k <= ep && !get_bit(labpos, k);
                    int tot;
                    const int r = block_rank(S, keep, tot);
                    if (keep) lf[nl + r] = cloud[k];
                    nl += tot;
                }
                if (threadIdx.x == 0) s_nl = nl;
                __syncthreads();
            }
            const int nl = s_nl;
            __syncthreads();
            block_voxel_grid(S, lf, nl, 0.2f, o_lflat, D.lf_cap, &S.counts[4], lds_keys, gk, D.key_cap);  // SR:667-673
            __syncthreads();
            if (threadIdx.x == 0 && ring_counts) {
                ring_counts[32 * b + ring] += S.counts[2];
                ring_counts[32 * b + 16 + ring] += S.counts[4];
            }
        }
    }
    if (threadIdx.x == 0) {
        S.counts[0] = cs;
        for (int k = 0; k < 5; ++k) counts[5 * b + k] = S.counts[k];
        if (S.overflow) atomicOr(status, 1);
    }
}

}  // namespace gpscal

using namespace gpscal;

static size_t sr_dyn_lds() { return sizeof(unsigned long long) * LDS_KEYS + sizeof(int) * LDS_KEYS + sizeof(unsigned) * 3 * SR_BIT_WORDS; }

namespace gpscal {

int scan_registration_device(gpscal_ctx *ctx, int nsweeps, const int *xyz_off, const int *lfo, const float *d_xyz,
                             float4 *d_full, float4 *d_sharp, float4 *d_lsharp, float4 *d_flat, float4 *d_lflat,
                             int *d_counts, int *status, int *d_ring_counts)
{
    std::vector<SrDesc> hd(nsweeps);
    long long key_total = 0;
    for (int b = 0; b < nsweeps; ++b) {
        SrDesc &D = hd[b];
        D.in_off = xyz_off[b];
        D.n_in = xyz_off[b + 1] - xyz_off[b];
        D.lf_off = lfo[b];
        D.lf_cap = lfo[b + 1] - lfo[b];
        if (D.n_in < 0 || D.lf_cap < 0) return fail(ctx, GPSCAL_EINVAL, "scan registration: bad offsets");
        int np2 = 1;
        while (np2 < D.n_in) np2 <<= 1;
        // sectors above LDS_KEYS points sort in HBM; twice the points: block_voxel_grid's radix sort wants a second buffer
        D.key_cap = np2 > LDS_KEYS ? std::max(np2, 2 * D.n_in) : 0;
        D.key_off = key_total;
        D.pad = 0;
        key_total += D.key_cap;
    }
    const size_t total = (size_t)std::max(xyz_off[nsweeps], 1);
    DevBuf<SrDesc> d_desc;
    DevBuf<signed char> d_ring;
    DevBuf<float> d_ori, d_curv;
    DevBuf<int> d_sid, d_old, d_status;
    DevBuf<float4> d_lfs;
    DevBuf<unsigned long long> d_keys;
    GPSCAL_HIP(ctx, d_desc.alloc_async(nsweeps, ctx->stream));
    GPSCAL_HIP(ctx, d_ring.alloc_async(total, ctx->stream));
    GPSCAL_HIP(ctx, d_ori.alloc_async(total, ctx->stream));
    GPSCAL_HIP(ctx, d_curv.alloc_async(total, ctx->stream));
    GPSCAL_HIP(ctx, d_sid.alloc_async(total, ctx->stream));
    GPSCAL_HIP(ctx, d_lfs.alloc_async(total, ctx->stream));
    GPSCAL_HIP(ctx, d_keys.alloc_async((size_t)key_total, ctx->stream));
    GPSCAL_HIP(ctx, d_old.alloc_async((size_t)key_total, ctx->stream));
    GPSCAL_HIP(ctx, d_status.alloc_async(1, ctx->stream));
    GPSCAL_HIP(ctx, hipMemsetAsync(d_status.p, 0, sizeof(int), ctx->stream));
    GPSCAL_HIP(ctx, hipMemcpyAsync(d_desc.p, hd.data(), sizeof(SrDesc) * nsweeps, hipMemcpyHostToDevice, ctx->stream));
    static bool attr_set = false;
    if (!attr_set) {
        GPSCAL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(scan_registration_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sr_dyn_lds()));
        attr_set = true;
    }
    hipLaunchKernelGGL(scan_registration_kernel, dim3(nsweeps), dim3(SBLOCK), sr_dyn_lds(), ctx->stream, d_desc.p,
                       d_xyz, d_full, d_sharp, d_lsharp, d_flat, d_lflat, d_ring.p, d_ori.p, d_curv.p, d_sid.p, d_lfs.p,
                       d_keys.p, d_old.p, d_counts, d_status.p, d_ring_counts);
    GPSCAL_HIP(ctx, hipGetLastError());
    GPSCAL_HIP(ctx, hipMemcpyAsync(status, d_status.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

}  // namespace gpscal

extern "C" int gpscal_scan_registration_batched(gpscal_ctx *ctx, int nsweeps, const float *xyz, const int *xyz_off,
                                                float *full_xyzi, float *sharp_xyzi, float *less_sharp_xyzi,
                                                float *flat_xyzi, float *less_flat_xyzi, const int *less_flat_off,
                                                int *counts)
{
    if (!ctx || nsweeps < 1 || !xyz || !xyz_off || !full_xyzi || !sharp_xyzi || !less_sharp_xyzi || !flat_xyzi ||
        !less_flat_xyzi || !counts)
        return fail(ctx, GPSCAL_EINVAL, "gpscal_scan_registration_batched: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    const int *lfo = less_flat_off ? less_flat_off : xyz_off;
    const size_t total = (size_t)std::max(xyz_off[nsweeps], 1), lf_total = (size_t)std::max(lfo[nsweeps], 1);
    InArg<float> a_in;
    OutArg<float> o_full, o_sh, o_ls, o_fl, o_lf;
    OutArg<int> o_cnt;
    GPSCAL_HIP(ctx, a_in.bind(ctx, xyz, total * 3));
    GPSCAL_HIP(ctx, o_full.bind(ctx, full_xyzi, total * 4));
    GPSCAL_HIP(ctx, o_sh.bind(ctx, sharp_xyzi, (size_t)nsweeps * 1536 * 4));
    GPSCAL_HIP(ctx, o_ls.bind(ctx, less_sharp_xyzi, (size_t)nsweeps * 1920 * 4));
    GPSCAL_HIP(ctx, o_fl.bind(ctx, flat_xyzi, (size_t)nsweeps * 3072 * 4));
    GPSCAL_HIP(ctx, o_lf.bind(ctx, less_flat_xyzi, lf_total * 4));
    GPSCAL_HIP(ctx, o_cnt.bind(ctx, counts, (size_t)nsweeps * 5));
    int st = 0;
    int rc = scan_registration_device(ctx, nsweeps, xyz_off, lfo, a_in.dev, reinterpret_cast<float4 *>(o_full.dev),
                                      reinterpret_cast<float4 *>(o_sh.dev), reinterpret_cast<float4 *>(o_ls.dev),
                                      reinterpret_cast<float4 *>(o_fl.dev), reinterpret_cast<float4 *>(o_lf.dev),
                                      o_cnt.dev, &st, nullptr);
    if (rc) return rc;
    bool sync = true;
    GPSCAL_HIP(ctx, o_full.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o_sh.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o_ls.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o_fl.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o_lf.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o_cnt.commit(ctx, &sync));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (st & 2) return fail(ctx, GPSCAL_ESIZE, "gpscal_scan_registration_batched: a sweep has more than 60000 ring points (POINTSNUM)");
    if (st & 1) return fail(ctx, GPSCAL_ERANGE, "gpscal_scan_registration_batched: less-flat output capacity exceeded");
    return GPSCAL_OK;
}

extern "C" int gpscal_voxel_grid_batched(gpscal_ctx *ctx, int nclouds, const float *pts_xyzi, const int *off,
                                         float leaf, float *out_xyzi, int *counts)
{
    if (!ctx || nclouds < 1 || !pts_xyzi || !off || !(leaf > 0.f) || !out_xyzi || !counts)
        return fail(ctx, GPSCAL_EINVAL, "gpscal_voxel_grid_batched: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<VgDesc> hd(nclouds);
    long long key_total = 0;
    for (int b = 0; b < nclouds; ++b) {
        VgDesc &D = hd[b];
        D.off = off[b];
        D.n = off[b + 1] - off[b];
        if (D.n < 0) return fail(ctx, GPSCAL_EINVAL, "gpscal_voxel_grid_batched: bad offsets");
        int np2 = 1;
        while (np2 < D.n) np2 <<= 1;
        D.key_cap = np2 > LDS_KEYS ? std::max(np2, 2 * D.n) : 0;  // room for the radix sort's second buffer
        D.key_off = key_total;
        key_total += D.key_cap;
    }
    const size_t total = (size_t)std::max(off[nclouds], 1);
    InArg<float> a_in;
    OutArg<float> o_out;
    OutArg<int> o_cnt;
    GPSCAL_HIP(ctx, a_in.bind(ctx, pts_xyzi, total * 4));
    GPSCAL_HIP(ctx, o_out.bind(ctx, out_xyzi, total * 4));
    GPSCAL_HIP(ctx, o_cnt.bind(ctx, counts, (size_t)nclouds));
    DevBuf<VgDesc> d_desc;
    DevBuf<unsigned long long> d_keys;
    DevBuf<int> d_status;
    GPSCAL_HIP(ctx, d_desc.alloc_async(nclouds, ctx->stream));
    GPSCAL_HIP(ctx, d_keys.alloc_async((size_t)key_total, ctx->stream));
    GPSCAL_HIP(ctx, d_status.alloc_async(1, ctx->stream));
    GPSCAL_HIP(ctx, hipMemsetAsync(d_status.p, 0, sizeof(int), ctx->stream));
    GPSCAL_HIP(ctx, hipMemcpyAsync(d_desc.p, hd.data(), sizeof(VgDesc) * nclouds, hipMemcpyHostToDevice, ctx->stream));
    const size_t lds = sizeof(unsigned long long) * LDS_KEYS;
    hipLaunchKernelGGL(voxel_grid_kernel, dim3(nclouds), dim3(SBLOCK), lds, ctx->stream, d_desc.p,
                       reinterpret_cast<const float4 *>(a_in.dev), leaf, reinterpret_cast<float4 *>(o_out.dev),
                       d_keys.p, o_cnt.dev, d_status.p);
    GPSCAL_HIP(ctx, hipGetLastError());
    bool sync = true;
    GPSCAL_HIP(ctx, o_out.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o_cnt.commit(ctx, &sync));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}
