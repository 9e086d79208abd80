// kf_integrate.hip - integrateKernel (cpp:628-673) and its launcher.
#include "kf_dev.h"

namespace kf {

// =====================================================================================
// integrateKernel (cpp:628-673).  A wave owns a 16 x 4 patch of (x, y) columns of one z-chunk (KFH_INT_TX; a block 32 x 8 columns):
// every voxel access is four 64-byte segments of the x-fastest short2 grid, and the compact footprint keeps the wave's lanes
// together in what they find (round 1 used 64 x 1: full 256-byte lines, but rows 1.2 m long that cut through everything).  The camera-space position is carried along z by
// repeated addition exactly like the CPU loop (cpp:645-646); a thread that starts at z0 > 0 first replays
// the z0 additions in registers so that its float state carries the bits of the serial march.
// The six running sums are kept as three float pairs and advanced with packed adds (v_pk_add_f32).
//
// Most voxels do not project into the image, and the kernel is bound by instruction issue, not by bytes (census and per-wave traces in
// profiles/r05_sweeps.md), so the work per rejected voxel - vector AND scalar instructions - is what matters.  The z march is cut
// into chunks (one wave per 16 x 4 patch and chunk) and the chunks into batches of IB voxels:
//   * chunk reject: the un-marched end voxels of the whole chunk against the image, with the launch's rounding bound to spare; a
//     thread whose chunk is out replays nothing and sits the batch loop out;
//   * batch reject: x/z and y/z are monotonic along the column while z stays positive, so if the batch's first voxel and the NEXT
//     batch's first voxel are outside the image on the same side with a quarter-pixel margin (four signed side distances per end
//     point, each end point evaluated once), or both are behind the camera, every voxel of the batch fails cpp:647-653 and only the
//     IB packed additions are spent on it;
//   * otherwise the detailed pass, straight-line code with selects: rcp-based projection, ONE near-integer test for both coordinates
//     that sends the wave to the two IEEE divisions, unconditional depth gathers (pixel 0 for lanes without a pixel), then the voxel
//     reads of the survivors;
//   * t = depth - cam.z decides cheaply what the reference computes through diff = t * sqrt(1 + (x/z)^2 + (y/z)^2):
//     since that factor is >= 1, t <= -mu means diff <= -mu (skip, no voxel access) and t >= mu means
//     diff / mu >= 1, i.e. sdf == 1.0f exactly; only the band |t| < mu needs the two divisions and the square root;
//   * free space that has only ever been free space (stored 32766, sdf 1) keeps its 32766 for every weight: only the weight moves.
// Every decision and every stored short is the one the CPU loop produces.
// =====================================================================================

typedef float v2f __attribute__((ext_vector_type(2)));

struct ColumnState { v2f a, b; float cy, cz; };   // (pos.x, pos.y), (pos.z, cam.x), cam.y, cam.z (cz unused when KZ)

#ifndef KFH_INT_BX
#define KFH_INT_BX 32
#endif
#ifndef KFH_INT_BY
#define KFH_INT_BY 8
#endif
#ifndef KFH_INT_TX
#define KFH_INT_TX 16
#endif
constexpr unsigned INT_BX = KFH_INT_BX, INT_BY = KFH_INT_BY, INT_TX = KFH_INT_TX;
static_assert(INT_BX * INT_BY == 256 && INT_BX % INT_TX == 0 && 64 % INT_TX == 0 && (INT_BX / INT_TX) * (INT_BY / (64 / INT_TX)) == 4,
              "a block of k_integrate is 256 columns = four waves of INT_TX x (64 / INT_TX)");

// KZ: the third row of K is (0, 0, 1, 0) (every pinhole matrix camera_matrix() builds; the launcher checks): then
// cam.z = 0*x + 0*y + 1*z + 0 and its step 0*dx + 0*dy + 1*dz are pos.z and its step bit for bit, the two running sums are the
// same sequence of floats, and the march carries five sums instead of six (a packed add becomes a plain one).
template <bool KZ> __device__ __forceinline__ void march_step(ColumnState &S, v2f da, v2f db, v2f dc) {
    S.a += da; S.b += db;
    S.cy += dc.x;
    if (!KZ) S.cz += dc.y;
}
template <bool KZ> __device__ __forceinline__ float cam_z(const ColumnState &S) { return KZ ? S.b.x : S.cz; }

// Stage probes for the instruction census (profiles/r05_integrate_census.txt): ONLY in a build with -DKFH_INT_PROBES=1 (through
// KFH_EXTRA_HIPCC_FLAGS) does the kernel look at `dbg` - results are wrong with a probe on; the shipped library has no such path
// (tests/test_capi_symbols.py greps the default build for the switch's name).
#ifndef KFH_INT_PROBES
#define KFH_INT_PROBES 0
#endif
#if KFH_INT_PROBES
#define KFH_INT_PROBE(bit) ((dbg & (bit)) != 0)
#else
#define KFH_INT_PROBE(bit) false
#endif
// (plain global stores: dirty in the XCD's L2 until the launch's closing write-back.  Round 5 measured the alternatives as raw buffer
// stores with aux bits, profiles/r05_ab_store_aux.txt: sc0 sc1 - written through as issued - 97.2 -> 103.0 us at 512^3, nt 97.9, aux 0 98.6.)
__device__ __forceinline__ void store_voxel(short2 *vol, unsigned byte_offset, unsigned value) {
    *reinterpret_cast<unsigned *>(reinterpret_cast<char *>(vol) + byte_offset) = value;
}
// Per-batch wall-clock stamps of every wave for the KFH_INTEGRATE_TRACE dump (profiles/int_stamps.py): ONLY in a build with
// -DKFH_INT_STAMPS=1; a record is then 32 words per wave - {start, end, work, block, after the head, after the replay, the start of each
// batch with the stage the previous batch reached in the top bits (1 detailed pass, 2 depth classification, 3 update), end of the loop}.
#ifndef KFH_INT_STAMPS
#define KFH_INT_STAMPS 0
#endif
constexpr unsigned INT_TRACE_WORDS = KFH_INT_STAMPS ? 32u : 4u;
#if KFH_INT_STAMPS
#define KFH_INT_STAMP(slot, tag)                                                                                               \
    do {                                                                                                                      \
        if (trace) {                                                                                                          \
            const unsigned long long t_ = wall_clock64() | ((unsigned long long)(tag) << 60);                                 \
            if ((unsigned)__builtin_amdgcn_readfirstlane((int)threadIdx.x) == threadIdx.x && (slot) < 32u)                     \
                trace[(size_t)INT_TRACE_WORDS * (slot_id * 4 + threadIdx.y) + (slot)] = t_;                                    \
        }                                                                                                                     \
    } while (0)
#else
#define KFH_INT_STAMP(slot, tag) do { } while (0)
#endif
#ifndef KFH_INTEGRATE_WAVES_PER_SIMD
#define KFH_INTEGRATE_WAVES_PER_SIMD 5      // occupancy floor handed to the register allocator (5 = at most 96 VGPRs)
#endif
template <int IB, bool KZ>
__global__ __launch_bounds__(256, IB <= 8 ? KFH_INTEGRATE_WAVES_PER_SIMD : 1) void k_integrate(short2 *__restrict__ vol, unsigned sx, unsigned sy, unsigned sz,
                                                   f3 dim, const float *__restrict__ depth, unsigned W, unsigned H,
                                                   const IntegrateParams *__restrict__ ip, float mu, float maxweight,
                                                   unsigned zchunk, unsigned long long *__restrict__ updated_total,
                                                   unsigned char *__restrict__ skip_flags, int dbg,
                                                   unsigned long long *__restrict__ trace,
                                                   const unsigned *__restrict__ order, unsigned *__restrict__ cost) {
    // everything a wave needs before it can decide anything - the launch parameters the control operation left in DevState
    // (gate, matrices, bounds) and its entry of the dispatch order - is requested in one go: written in the order of use
    // these are five dependent scalar round trips, ~1.5 us at the head of every wave, two thirds of which leave right after
    const unsigned slot = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
#if KFH_INT_STAMPS
    const unsigned slot_id = slot;
    unsigned stage_reached = 0;
#endif
    const IntegrateParams P = *ip;
    const unsigned vb = order ? order[slot] : slot;
    // (the empty statement "uses" one word of every piece the structure is fetched in, so that the compiler cannot put the
    // matrices' loads back behind the gate)
    asm volatile("" : : "s"(P.invTrack.m[0][0]), "s"(P.invTrack.m[2][0]), "s"(P.K.m[0][0]), "s"(P.K.m[2][0]), "s"(vb));
    if (!P.enabled) return;
    const unsigned long long t_start = trace ? wall_clock64() : 0ull;
    // Launch slot -> block of the (x / 64, y / 4, z-chunk) grid.  As in k_raycast: the launch ends with its slowest
    // wave (a column chunk lying in the truncation band costs ~10x one outside the frustum), so the blocks that were
    // expensive in the previous launch are dispatched first instead of wherever x-y-z order puts them (at 512^3 the
    // far chunks, i.e. the band, used to start 55 us into a 170 us launch).
    const unsigned vbx = vb % gridDim.x, vby = (vb / gridDim.x) % gridDim.y, vbz = vb / (gridDim.x * gridDim.y);
    // Columns of a block: INT_BX x INT_BY; of a wave: INT_TX x (64 / INT_TX), the waves side by side in x first.  A compact
    // wave footprint keeps the 64 lanes together in what they find - inside or outside the frustum, in front of or behind
    // the surface - and a row of INT_TX voxels is still a 64-byte segment at 16 (profiles/r02_sweeps.md).
    const unsigned wave_ = threadIdx.y, lane_ = threadIdx.x;
    constexpr unsigned WAVES_X = INT_BX / INT_TX, TILE_Y = 64 / INT_TX;
    const unsigned x = vbx * INT_BX + (wave_ % WAVES_X) * INT_TX + lane_ % INT_TX;
    const unsigned y = vby * INT_BY + (wave_ / WAVES_X) * TILE_Y + lane_ / INT_TX;
    unsigned updated = 0, work = 0;
    if (x < sx && y < sy) {
        const M4 invTrack = P.invTrack, K = P.K;
        const f3 delta = rotate(invTrack, mk3(0.f, 0.f, dim.z / sz));
        const f3 cameraDelta = rotate(K, delta);
        const f3 pos0 = xform(invTrack, mk3((x + 0.5f) * dim.x / sx, (y + 0.5f) * dim.y / sy, (0 + 0.5f) * dim.z / sz));
        const f3 cam0 = xform(K, pos0);
        ColumnState S;
        S.a = (v2f){pos0.x, pos0.y}; S.b = (v2f){pos0.z, cam0.x}; S.cy = cam0.y; S.cz = cam0.z;
        const v2f da = (v2f){delta.x, delta.y}, db = (v2f){delta.z, cameraDelta.x}, dc = (v2f){cameraDelta.y, cameraDelta.z};
        const unsigned z0 = vbz * zchunk;
        const unsigned z1 = z0 + zchunk < sz ? z0 + zchunk : sz;
        // Whole-chunk reject BEFORE the replay.  The marched state of voxel z is the exact line pos0 + z * delta plus at
        // most z rounded additions, each off by <= 2^-24 of the value: |error| <= E = (sz + 4) 2^-24 cam_bound for cam.x / cam.y
        // and <= Ez = (sz + 4) 2^-24 far_bound for cam.z = pos.z.  The reference skips a voxel when pos.z < 1e-4 or the pixel
        // cam.x / cam.z + 0.5 falls outside [0, W-1] x [0, H-1] (cpp:647-653); "outside on the left" is cam.x + 0.5 cam.z < 0
        // for cam.z > 0, a LINEAR condition, so if the un-marched end points of the chunk satisfy it with E + 0.5 Ez to
        // spare (plus a quarter pixel for the rounding of the reference's own division), every marched voxel in between
        // does: the chunk updates nothing and the thread leaves without replaying its z0 additions.  Four of five chunks
        // lie outside the frustum; what remains pays the replay, which is what lets the chunks be short (see integrate_grid).
        bool chunk_out = false;
        if (!KFH_INT_PROBE(8)) {
            const float zf = (float)z0, zl = (float)(z1 - 1);
            const float n_err = (float)(sz + 4) * 5.9604645e-8f;
            const float E = n_err * P.cam_bound, Ez = n_err * P.far_bound;
            const float pzf = __builtin_fmaf(zf, delta.z, pos0.z), pzl = __builtin_fmaf(zl, delta.z, pos0.z);
            const float cxf = __builtin_fmaf(zf, cameraDelta.x, cam0.x), cxl = __builtin_fmaf(zl, cameraDelta.x, cam0.x);
            const float cyf = __builtin_fmaf(zf, cameraDelta.y, cam0.y), cyl = __builtin_fmaf(zl, cameraDelta.y, cam0.y);
            const float czf = __builtin_fmaf(zf, cameraDelta.z, cam0.z), czl = __builtin_fmaf(zl, cameraDelta.z, cam0.z);
            const bool behind = pzf < -Ez - 1e-3f && pzl < -Ez - 1e-3f;           // every marched pos.z < 1e-4
            if (czf > Ez + 1e-3f && czl > Ez + 1e-3f) {                            // every marched cam.z > 0
                const float wq = (float)W - 0.25f, hq = (float)H - 0.25f;         // right / bottom: pixel > size - 1, i.e. c > (size - 0.5) z
                const float el = E + 0.75f * Ez, er = E + wq * Ez, eb = E + hq * Ez;
                const bool left = cxf + 0.75f * czf < -el && cxl + 0.75f * czl < -el;
                const bool right = cxf - wq * czf > er && cxl - wq * czl > er;
                const bool top = cyf + 0.75f * czf < -el && cyl + 0.75f * czl < -el;
                const bool bottom = cyf - hq * czf > eb && cyl - hq * czl > eb;
                chunk_out = left || right || top || bottom;
            }
            chunk_out = chunk_out || behind;
        }
        KFH_INT_STAMP(4u, 0);
        if (!chunk_out) {      // (eight steps per trip: as a plain loop every step pays the loop's scalar bookkeeping too)
            unsigned z = 0;
            for (; z + 8 <= z0; z += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) march_step<KZ>(S, da, db, dc);
            }
            for (; z < z0; ++z) march_step<KZ>(S, da, db, dc);
        }
        const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
        // per-voxel test: the very floats the reference divides, 0.1 px of margin covers the rounding of the product.
        // batch test: interior voxels are the endpoints' interpolation plus at most IB roundings of the running sums, each
        // <= 2^-24 |cam.x|; with cam.z > 0.05 that moves x/z by IB * 2^-24 * max|cam.x| / 0.05 pixels, which must stay below
        // 0.2 of the 0.25 px margin.  max|cam.x| over the volume is bounded per launch (IntegrateParams::cam_bound, from the
        // intrinsics and the camera position): where the bound fails - huge intrinsics or volumes - no batch is rejected
        // on its end voxels and every voxel takes the reference tests.
        const bool batch_ok = (float)(IB + 1) * 5.9604645e-8f * P.cam_bound < 0.2f * 0.05f;
        const float wlim_b = (float)W - 0.25f, hlim_b = (float)H - 0.25f;
        const float ptol = 2e-6f * (float)(W > H ? W : H);
        // (voxel byte offsets in 32 bits: the launcher refuses volumes above 2^30 voxels, sizes_addressable)
        const unsigned plane4 = sx * sy * 4u;
        unsigned voff = (x + y * sx) * 4u + z0 * plane4;
        const float wclamp = (float)W + 0.5f, hclamp = (float)H + 0.5f;
        // (the batch counter is the same in every lane - a lane whose chunk is out sits the loop out - so that "is voxel i part of
        // this batch" is a scalar question)
        // End-voxel test of a batch: the four signed distances of an end point from the image's sides - negative = outside on that
        // side, with the quarter-pixel margin - are kept from one batch to the next: a batch is judged on its first voxel and the
        // first voxel of the NEXT batch (one voxel more than it holds, which the bound above allows for), so every end point is
        // evaluated once; "both outside on the same side" is max(first, next) < 0 for some side.
        struct EndPoint { float l, r, t, b, cz, z; };
        auto end_point = [&](const ColumnState &E) {
            const float cz = cam_z<KZ>(E);
            EndPoint e;
            e.l = __builtin_fmaf(0.75f, cz, E.b.y); e.r = __builtin_fmaf(wlim_b, cz, -E.b.y);
            e.t = __builtin_fmaf(0.75f, cz, E.cy); e.b = __builtin_fmaf(hlim_b, cz, -E.cy);
            e.cz = cz; e.z = E.b.x;
            return e;
        };
        EndPoint ep = end_point(S);
        KFH_INT_STAMP(5u, 0);
        if (!chunk_out)
        for (unsigned zb = z0; zb < z1; zb += IB, voff += (unsigned)IB * plane4) {
            const unsigned nb = (z1 - zb) < (unsigned)IB ? (z1 - zb) : (unsigned)IB;
#if KFH_INT_STAMPS
            {
                const unsigned reached = __builtin_amdgcn_ballot_w64(stage_reached >= 3u) ? 3u : (__builtin_amdgcn_ballot_w64(stage_reached >= 2u) ? 2u : (__builtin_amdgcn_ballot_w64(stage_reached >= 1u) ? 1u : 0u));
                KFH_INT_STAMP(6u + (zb - z0) / (unsigned)IB, reached);
                stage_reached = 0;
            }
#endif
            const ColumnState first = S;
            // advance over the batch: S becomes the state of the next batch's first voxel
            if (nb == (unsigned)IB) {   // whole batch (all but possibly the last of a chunk): straight-line packed adds
#pragma unroll
                for (int i = 0; i < IB; ++i) { march_step<KZ>(S, da, db, dc); }
            } else {
                for (unsigned i = 0; i < nb; ++i) march_step<KZ>(S, da, db, dc);
            }
            {
                const EndPoint en = end_point(S);
                const bool behind = __builtin_fmaxf(ep.z, en.z) < 0.00005f;
                const float side = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ep.l, en.l), __builtin_fmaxf(ep.r, en.r)),
                                                   __builtin_fminf(__builtin_fmaxf(ep.t, en.t), __builtin_fmaxf(ep.b, en.b)));
                const bool outside = batch_ok && __builtin_fminf(ep.cz, en.cz) > 0.05f && side < 0.f;
                ep = en;
                if (behind || outside || KFH_INT_PROBE(1)) continue;
            }
            // ---- detailed pass over the batch, replaying the additions from its first voxel: straight-line code with selects.  The
            // heaviest waves - every batch in the frustum and near the surface - are what the launch waits for, and a lone wave issues one
            // instruction per ~8 cycles whatever its kind: round 4's form of this pass spent ~20 scalar instructions per voxel on nested
            // exec-mask branches next to ~35 vector ones (profiles/r05_integrate_census.txt).
            // cpp:649-654 needs only the comparisons with 0 / W-1 / H-1 and the integer part of fl(fl(c/z) + 0.5).  c * rcp(z) is within
            // 3 ulp of fl(c/z), i.e. within 3.6e-7 * max(W,H) pixels wherever those outcomes are undecided; unless it lands within `ptol`
            // (5x that) of an integer it gives the same decisions and the same integer; the IEEE divisions run only for those rare
            // pixels (a wave-level branch).  Values far outside are clamped to -1.5 / size + 0.5 first - half-integers, so never "near an
            // integer", and on the same side of 0 and size - 1 as the value itself.
#if KFH_INT_STAMPS
            stage_reached = 1;
#endif
            ColumnState T = first;
            unsigned live = 0;            // bit i: voxel zb+i projects into the image
            unsigned pix[IB];             // byte offset of its pixel in the depth image
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                if ((unsigned)i < nb) {
                    const float posz = T.b.x, cx = T.b.y, cy = T.cy, cz = cam_z<KZ>(T);
                    const bool okz = KZ ? !(posz < 0.0001f) : (!(posz < 0.0001f) && cz > 0.f);      // cpp:647 (cam.z = pos.z > 0 when KZ)
                    const float rz = __builtin_amdgcn_rcpf(cz);
                    float px = __builtin_amdgcn_fmed3f(cx * rz + 0.5f, -1.5f, wclamp), py = __builtin_amdgcn_fmed3f(cy * rz + 0.5f, -1.5f, hclamp);
                    const float near_int = __builtin_fminf(fabsf(px - rintf(px)), fabsf(py - rintf(py)));
                    if (okz && !(near_int >= ptol)) {          // (also a NaN: cam.z = 0 with K's third row not (0, 0, 1, 0))
                        px = cx / cz + 0.5f;
                        py = cy / cz + 0.5f;
                    }
                    const bool inside = okz && !(px < 0 || px > wm1 || py < 0 || py > hm1);        // cpp:651-653
                    live |= inside ? (1u << i) : 0u;
                    // (pixel 0 for the others: a load nobody looks at; 24-bit multiply: the launcher refuses images of 2^26 pixels and more)
                    pix[i] = inside ? __umul24((unsigned)py, W * 4u) + ((unsigned)px << 2) : 0u;
                    march_step<KZ>(T, da, db, dc);
                } else {
                    pix[i] = 0u;      // (a voxel beyond the chunk's end: its depth gather below is unconditional and must have an address)
                }
            }
            ++work;
            if (!live || KFH_INT_PROBE(2)) continue;
            // t = depth - cam.z: <= -mu -> rejected by cpp:662; >= mu -> sdf == 1; in between -> full formula
#if KFH_INT_STAMPS
            stage_reached = 2;
#endif
            unsigned need = 0;
            float t_[IB];
            {
                float d[IB];
#pragma unroll
                for (int i = 0; i < IB; ++i) d[i] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(depth) + pix[i]);
                float camz_i = cam_z<KZ>(first);  // cam.z of voxel i, replayed like the positions below
                const float dz_step = KZ ? db.x : dc.y;
#pragma unroll
                for (int i = 0; i < IB; ++i, camz_i += dz_step) {
                    const float t = d[i] - camz_i;
                    t_[i] = t;
                    need |= ((live & (1u << i)) && d[i] != 0 && t > -mu) ? (1u << i) : 0u;        // cpp:655, cpp:662 (diff <= t for t < 0)
                }
            }
            if (!need || KFH_INT_PROBE(4)) continue;
            work += 4;
#if KFH_INT_STAMPS
            stage_reached = 3;
#endif
            unsigned c[IB];               // the voxels as stored: tsdf short in the low half, weight in the high half
#pragma unroll
            for (int i = 0; i < IB; ++i)
                if (need & (1u << i)) c[i] = *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(vol) + (voff + (unsigned)i * plane4));
#if KFH_INT_PROBES
            if (KFH_INT_PROBE(16)) {      // (the loads stay: each fetched voxel is "used")
#pragma unroll
                for (int i = 0; i < IB; ++i)
                    if (need & (1u << i)) asm volatile("" : : "v"(c[i]));
                continue;
            }
#endif
            // the world position of a band voxel is replayed from the batch's first state instead of being kept in 24 registers across
            // the two memory round trips above
            v2f Rxy = first.a;
            float Rz = first.b.x;
#pragma unroll
            for (int i = 0; i < IB; ++i, Rxy += da, Rz += db.x) {
                if (!(need & (1u << i))) continue;
                const short cx16 = (short)(c[i] & 0xFFFFu), cw16 = (short)(c[i] >> 16);
                // Free space in front of the surface that has only ever been free space: 32766 * kShort2Float is 1.0f exactly, so the
                // running average (w * 1 + 1) / (w + 1) is 1 and the stored short 32766 again, for every weight; only the weight moves
                // (the same float operations as below), and a saturated voxel is stored back unchanged - i.e. not at all.
                // (Taken when every lane of the wave that updates voxel i is of that kind - the usual case away from the surface;
                // otherwise all of them go through the formula, which gives those lanes the same bits.)
                if (__builtin_amdgcn_ballot_w64(!(t_[i] >= mu && cx16 == 32766)) == 0ull) {
                    const short w16 = (short)fmin_h((float)cw16 + 1, maxweight);
                    if (w16 != cw16) store_voxel(vol, voff + (unsigned)i * plane4, 32766u | ((unsigned)(unsigned short)w16 << 16));
                    ++updated;
                    continue;
                }
                float sdf;
                if (t_[i] >= mu) {
                    sdf = 1.f;
                } else {
                    const float rx = Rxy.x, ry = Rxy.y, rz = Rz;
                    const float diff = t_[i] * sqrtf(1 + sq(rx / rz) + sq(ry / rz));
                    if (!(diff > -mu)) continue;
                    sdf = fmin_h(1.f, diff / mu);
                }
                float tsdf = cx16 * kShort2Float, w = cw16;
                tsdf = clamp_h((w * tsdf + sdf) / (w + 1), -1.f, 1.f);
                w = fmin_h(w + 1, maxweight);
                const short ts = (short)(tsdf * 32766.0f);
                store_voxel(vol, voff + (unsigned)i * plane4, (unsigned)(unsigned short)ts | ((unsigned)(unsigned short)(short)w << 16));
                if (skip_flags && ts < kSkipShortFwd && cx16 >= kSkipShortFwd)   // flags are sticky: first crossing only
                    mark_bricks(skip_flags, bricks_along(sx), bricks_along(sy), bricks_along(sz), x, y, zb + i);
                ++updated;
            }
        }
    }
#if KFH_INT_STAMPS
    {
        const unsigned reached = __builtin_amdgcn_ballot_w64(stage_reached >= 3u) ? 3u : (__builtin_amdgcn_ballot_w64(stage_reached >= 2u) ? 2u : (__builtin_amdgcn_ballot_w64(stage_reached >= 1u) ? 1u : 0u));
        KFH_INT_STAMP(31u, reached);
    }
#endif
    // two of three waves examined nothing (their chunks lie outside the frustum): nothing to add to the cost table or the
    // counter, so they leave here instead of running two six-step wave reductions over zeros
    if (!trace && __builtin_amdgcn_ballot_w64((work | updated) != 0u) == 0ull) return;
    if (cost || trace) {   // cost of the block = the busiest lane of its waves: batches examined + 4 x batches updated
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) work = umax_(work, (unsigned)__shfl_xor((int)work, off, 64));
        if (cost && threadIdx.x == 0 && work) atomicMax(cost + vb, work);
    }
    if (trace && threadIdx.x == 0) {   // debugging (KFH_INTEGRATE_TRACE): per wave {start, end, cost, block}
        const size_t wave = (size_t)INT_TRACE_WORDS * (slot * 4 + threadIdx.y);
        trace[wave] = t_start; trace[wave + 1] = wall_clock64(); trace[wave + 2] = work; trace[wave + 3] = vb;
    }
    // instrumentation (U = voxels that passed all four tests): one atomic per wave
    if (updated_total) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) updated += __shfl_down(updated, off, 64);
        if ((threadIdx.x & 63) == 0 && updated) atomicAdd(updated_total, (unsigned long long)updated);
    }
}

namespace launch {

// enough z-chunks to put >= ~8192 waves in flight on 256 CUs (each chunk replays its z0 additions, so not more: at
// 512^3 four chunks of 128 are the measured optimum)
static dim3 integrate_grid(unsigned sx, unsigned sy, unsigned sz, unsigned *zchunk_out) {
    static const long chunks_env = env_long("KFH_INTEGRATE_CHUNKS", 1, 64, 0);
    const unsigned waves_xy = ((sx + 63) / 64) * sy;     // (waves of 64 columns, whatever their shape)
    unsigned nchunks = chunks_env ? (unsigned)chunks_env : (8192 + waves_xy - 1) / waves_xy;
    if (!chunks_env && nchunks < 4 && sz >= 256) nchunks = 4;
    if (nchunks < 1) nchunks = 1;
    if (!chunks_env && nchunks > 16) nchunks = 16;
    if (nchunks > sz) nchunks = sz;
    const unsigned zchunk = (sz + nchunks - 1) / nchunks;
    if (zchunk_out) *zchunk_out = zchunk;
    return dim3((sx + INT_BX - 1) / INT_BX, (sy + INT_BY - 1) / INT_BY, (sz + zchunk - 1) / zchunk);
}
unsigned integrate_blocks(unsigned sx, unsigned sy, unsigned sz) { const dim3 g = integrate_grid(sx, sy, sz, nullptr); return g.x * g.y * g.z; }
bool integrate_reorder_enabled() { static const bool on = [] { const char *e = getenv("KFH_INTEGRATE_REORDER"); return !(e && e[0] == '0'); }(); return on; }
void integrate_order(unsigned *block_order, unsigned *block_cost, unsigned n, hipStream_t s) {   // sort as a launch of its own
    dispatch_order(block_order, block_cost, n, 1u, 1u, s);
}

void integrate(int16_t *vol, unsigned sx, unsigned sy, unsigned sz, f3 dim, const float *depth, unsigned W, unsigned H,
               const IntegrateParams *p, float mu, float maxweight, unsigned long long *updated_total, unsigned char *skip_flags,
               unsigned *block_order, unsigned *block_cost, bool k_trivial, hipStream_t s) {
    if (!sizes_addressable(sx, sy, sz, W, H)) { refused("integrate: volume above 2^30 voxels (or an x-y plane of 2^24) or an image of 2^26 pixels: outside the kernel's 32-bit addressing"); return; }
    static const long batch_env = env_long("KFH_INTEGRATE_BATCH", 1, 16, 8);
#if KFH_INT_PROBES
    // census build only (-DKFH_INT_PROBES=1; results are wrong with a probe on): 1 every batch rejected, 2 stop after the projection
    // pass, 4 stop after the depth classification, 16 stop after the voxel loads, 8 no whole-chunk reject; KFH_INT_PROBE_FROM = first launch they apply to
    static const int dbg_env = (int)env_long("KFH_INT_PROBE_STAGE", 0, 31, 0);
    static const long dbg_from = env_long("KFH_INT_PROBE_FROM", 0, 1000000, 0);
    static std::atomic<long> n_launch{0};
    const int dbg = (dbg_env && n_launch.fetch_add(1) >= dbg_from) ? dbg_env : 0;
#else
    const int dbg = 0;
#endif
    unsigned zchunk;
    const dim3 grid = integrate_grid(sx, sy, sz, &zchunk);
    short2 *v = reinterpret_cast<short2 *>(vol);
    static WaveTrace wt("KFH_INTEGRATE_TRACE");
    unsigned long long *trace = wt.arm((size_t)grid.x * grid.y * grid.z * 4 * (INT_TRACE_WORDS / 4), s);
    const bool reorder = integrate_reorder_enabled() && block_order && block_cost;
    const unsigned *order = reorder ? block_order : nullptr;
    unsigned *cost = reorder ? block_cost : nullptr;
    // (k_trivial: the third row of K is (0, 0, 1, 0), see KZ)
#define KFH_LAUNCH_INTEGRATE(IB_)                                                                                        \
    do {                                                                                                                 \
        if (k_trivial) k_integrate<IB_, true><<<grid, block_rows(), 0, s>>>(v, sx, sy, sz, dim, depth, W, H, p, mu, maxweight, zchunk, updated_total, skip_flags, dbg, trace, order, cost);  \
        else k_integrate<IB_, false><<<grid, block_rows(), 0, s>>>(v, sx, sy, sz, dim, depth, W, H, p, mu, maxweight, zchunk, updated_total, skip_flags, dbg, trace, order, cost);           \
    } while (0)
    if (batch_env >= 16) KFH_LAUNCH_INTEGRATE(16);
    else if (batch_env >= 8) KFH_LAUNCH_INTEGRATE(8);
    else if (batch_env >= 4) KFH_LAUNCH_INTEGRATE(4);
    else KFH_LAUNCH_INTEGRATE(1);
#undef KFH_LAUNCH_INTEGRATE
    launched("k_integrate");
    wt.dump();
}

}  // namespace launch
}  // namespace kf
