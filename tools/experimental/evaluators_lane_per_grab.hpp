// evaluators_lane_per_grab.hpp -- EXPERIMENTAL, not part of libhpmvs_amd.so.
// Two measured-slower forms of the photometric evaluator (one lane per grab: wave_eval_jobs, wave_eval_compact),
// moved out of hpmvs_amd/csrc/photometric.hpp in round 3.  They were byte-identical to the product
// evaluator; C1: 46.8 ms vs 38.2 ms (profiles/r02/evaluator_variants_c1.txt).  Kept for the record; to build them
// again, include this file after photometric.hpp inside a kernel TU.
#pragma once
namespace hpmvs {
// ------------------------------------------------------------------------------------------
// LANE-PER-GRAB exact-order evaluator: every (request, image) pair is one lane.
//
// The packed evaluator above spreads ONE grab over 49 lanes and has to hand the reference's sequential float32
// sums to single lanes through LDS (5 barriers and ~2.1 k wave-cycles per grab).  Here a lane does a whole
// sampleTexture by itself, exactly like the CPU thread of the reference does: the 49 bilinear taps in the
// reference's incremental stepping order (PatchOptimizer.cpp:511-526), the 147 raw values in registers,
// Patch2d::normalize's three sequential passes (Patch2d.hpp:46-84) and the 147-term dot product
// (Patch2d.hpp:37-44) with the request's reference texture, whose normalised values are read from the lane that
// grabbed the reference image with ds_bpermute (no LDS allocation, no barriers).  Up to 64 grabs advance per
// pass, the sums need no staging, and every float operation is the reference's in the reference's order.
// Requests are packed into passes like in wave_eval_packed (whole requests, <= 64 jobs per pass).
__device__ __forceinline__ void grab_window(const Prep& P, float (&t)[3 * kTexPix]) {
    float lx = P.tlx, ly = P.tly;  // row start: l += dy per row, c += dx per column
#pragma unroll
    for (int yy = 0; yy < kTex; yy++) {
        float cx = lx, cy = ly;
#pragma unroll
        for (int xx = 0; xx < kTex; xx++) {
            const int ix = (int)cx, iy = (int)cy;
            const uint8_t* p0 = P.pix + 3 * (iy * P.W + ix);
            Taps tp;
            tp.a = load8(p0);
            tp.b = load8(p0 + 3 * P.W);
            const float dx1 = cx - (float)ix, dx0 = 1.0f - dx1;
            const float dy1 = cy - (float)iy, dy0 = 1.0f - dy1;
            tp.f00 = dx0 * dy0; tp.f01 = dx0 * dy1; tp.f10 = dx1 * dy0; tp.f11 = dx1 * dy1;
            const int k = 3 * (yy * kTex + xx);
            texel(tp, t[k], t[k + 1], t[k + 2]);
            cx += P.dxx; cy += P.dxy;
        }
        lx += P.dyx; ly += P.dyy;
        // one row of taps (14 loads) in flight at a time: without this fence the scheduler hoists all 98 loads
        // of the window and spills the texture to scratch
        asm volatile("" ::: "memory");
    }
}

// Patch2d::normalize (Patch2d.hpp:46-84) on a register-resident texture
__device__ __forceinline__ void normalize_window(float (&t)[3 * kTexPix]) {
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < kTexPix; i++) { s0 += t[3 * i]; s1 += t[3 * i + 1]; s2 += t[3 * i + 2]; }
    const float a0 = div_by(s0, 1.0 / 49.0), a1 = div_by(s1, 1.0 / 49.0), a2 = div_by(s2, 1.0 / 49.0);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < kTexPix; i++) {
        const float f0 = a0 - t[3 * i], f1 = a1 - t[3 * i + 1], f2 = a2 - t[3 * i + 2];
        q += f0 * f0 + f1 * f1 + f2 * f2;
    }
    float v2 = sqrtf(div_by(q, 1.0 / 147.0));
    if (v2 == 0.0f) v2 = 1.0f;
    const double iv = recip64(v2);
#pragma unroll
    for (int i = 0; i < kTexPix; i++) {
        t[3 * i] = div_by(t[3 * i] - a0, iv);
        t[3 * i + 1] = div_by(t[3 * i + 1] - a1, iv);
        t[3 * i + 2] = div_by(t[3 * i + 2] - a2, iv);
    }
}

// myrow: the calling lane's own image list (uint16 ids; only read for lanes with a pending request).
// Must be called by all 64 lanes (wave-uniform control flow around every cross-lane operation).
#ifndef HPMVS_EVAL_JOBS_INLINE
#define HPMVS_EVAL_JOBS_ATTR __noinline__
#else
#define HPMVS_EVAL_JOBS_ATTR __forceinline__
#endif
// (out of line by default: its ~200 VGPRs -- the 147 texture values -- then do not compete with the caller's
//  long-lived per-patch registers inside the tap loop; the arguments travel by value, in registers)
template <bool COHERENT>
__device__ HPMVS_EVAL_JOBS_ATTR PackedRes wave_eval_jobs(DevScene sc, DevOptions o, const uint16_t* myrow,
                                                       unsigned long long pending, PackedReq q, PackedRes out) {
    const int lane = lane_id();
    const int q_refsel = q.obj_mode ? 0 : q.ref;
    const unsigned long long rowbits = (unsigned long long)myrow;
    const int row_lo = (int)(rowbits & 0xffffffffull), row_hi = (int)(rowbits >> 32);

    unsigned long long rem = pending;
    while (rem) {
        // ---- a pass: whole requests whose jobs fit the 64 lanes (at least one request)
        unsigned long long batch = 0ull;
        int J = 0;
        for (unsigned long long t = rem; t; t &= t - 1ull) {
            const int p = __ffsll((long long)t) - 1;
            const int np = bcast_i(q.nimg, p);
            if (batch && J + np > 64) break;
            batch |= 1ull << p;
            J += np;
        }
        rem &= ~batch;
        PH_T(t1);
        // ---- job of this lane (request myp, image index myi); each requesting lane learns where its jobs sit
        int myp = -1, myi = 0, refl = 0, myoff = 0, maxnp = 0;
        {
            int off = 0;
            for (unsigned long long t = batch; t; t &= t - 1ull) {
                const int p = __ffsll((long long)t) - 1;
                const int np = bcast_i(q.nimg, p);
                const int refp = bcast_i(q_refsel, p);
                if (lane >= off && lane < off + np) { myp = p; myi = lane - off; refl = off + refp; }
                if (lane == p) myoff = off;
                maxnp = np > maxnp ? np : maxnp;
                off += np;
            }
        }
        const int src = myp >= 0 ? myp : 0;
        float c[3], zax[3], pxa[3], pya[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            c[k] = shfl_f(q.c[k], src); zax[k] = shfl_f(q.zax[k], src);
            pxa[k] = shfl_f(q.px[k], src); pya[k] = shfl_f(q.py[k], src);
        }
        const float scale = shfl_f(q.scale, src);
        const unsigned long long rb = ((unsigned long long)(unsigned int)shfl_i(row_hi, src) << 32) |
                                      (unsigned long long)(unsigned int)shfl_i(row_lo, src);
        Prep P;
        P.ok = 0; P.tlx = P.tly = P.dxx = P.dxy = P.dyx = P.dyy = 0.0f; P.pix = nullptr; P.W = 0;
        if (myp >= 0) P = prep_image(sc.views[list_id<COHERENT>((const uint16_t*)rb, myi)], o, c, scale, pxa, pya, zax);
        const unsigned long long okmask = __ballot(P.ok != 0);
        PH_T(t2);
        PH_ADD(t1, t2, 13);

        // ---- the grab: sampleTexture + normalize, one lane per (request, image) whose reference image is usable
        const bool refok = (okmask >> refl) & 1ull;
        const bool work = myp >= 0 && P.ok && refok;
        float t[3 * kTexPix];
#pragma unroll
        for (int k = 0; k < 3 * kTexPix; k++) t[k] = 0.0f;
        if (work) {
            grab_window(P, t);
            normalize_window(t);
        }
        PH_T(t3);
        PH_ADD(t2, t3, 14);
        // ---- Patch2d::dot with the reference texture (held by lane refl)
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 3 * kTexPix; k++) acc += shfl_f(t[k], refl) * t[k];
        const float raw = 1.0f - div_by(acc, 1.0 / 147.0);
        const float rob = robustincc(raw);
        PH_T(t4);
        PH_ADD(t3, t4, 21);

        // ---- delivery: every requesting lane walks its own jobs in image order
        const bool mine = (batch >> lane) & 1ull;
        const int np = mine ? q.nimg : 0;
        const bool myrefok = mine && np > 0 && ((okmask >> (myoff + q_refsel)) & 1ull);
        double val = 0.0;
        float isum = 0.0f;
        unsigned long long keep = 0ull;
        int nok = 0, ngr = myrefok ? 1 : 0;
        for (int i = 0; i < maxnp; i++) {
            const int jl = (myoff + i) & 63;
            const float e_raw = shfl_f(raw, jl), e_rob = shfl_f(rob, jl);
            if (i < np) {
                const bool jok = myrefok && ((okmask >> jl) & 1ull);
                if (q.obj_mode) {
                    if (jok && i != q_refsel) { val += (double)e_rob; nok++; ngr++; }
                } else {
                    float e = 2.0f;
                    if (myrefok && i == q_refsel) e = 0.0f;
                    else if (jok) { e = q.robust ? e_rob : e_raw; ngr++; }
                    isum += e;
                    if (e < 1.0f - q.thr) keep |= 1ull << i;
                }
            }
        }
        if (mine) {
            out.ngrabs += ngr;
            if (q.obj_mode) out.obj = (!myrefok || nok < o.MIN_IMAGES_PER_PATCH - 1) ? 2.0 : val / (double)nok;
            else { out.keep = myrefok ? keep : 0ull; out.incc_sum = isum; }
        }
        PH_T(t5);
        PH_ADD(t4, t5, 22);
    }
    return out;
}

// ------------------------------------------------------------------------------------------
// Lane-per-grab evaluator with COMPACTION, for wavefronts that hold many requests (the stream kernel's EVAL visit:
// 64 requests of 64 different patches).  wave_eval_jobs gives a lane to every attached image, usable or not; with
// the long image lists after addImages most of them fail sampleTexture's gates (angle, level, window inside the
// image) and their lanes idle through the 5 k-instruction grab.  Here
//   phase 1  one lane per (request, image) runs ONLY the gates (prep_image, ~300 instructions): every request gets
//            the bit mask of its usable images;
//   phase 2  whole requests are packed by their USABLE image count, <= 64 grabs per pass, every lane of a pass
//            does a real grab (gates recomputed for the chosen image: cheaper than carrying 8 registers per job
//            through LDS); reference textures travel by ds_bpermute inside the pass as in wave_eval_jobs.
// Delivery walks each request's images in ascending index, unusable ones contributing the reference's constants
// (2.0 for a failed grab, PatchOptimizer.cpp:458-470), so every sum is formed in the reference's order.
#ifdef SK_STATS
__device__ unsigned long long* g_evdbg = nullptr;  // 16 counters, set by the stream kernel (stats builds only)
#define EV_T(v) const long long v = clock64()
#define EV_ACC(i, a, b) do { if (lane_id() == 0 && g_evdbg) atomicAdd(g_evdbg + (i), (unsigned long long)((b) - (a))); } while (0)
#else
#define EV_T(v) do { } while (0)
#define EV_ACC(i, a, b) do { } while (0)
#endif
template <bool COHERENT>
__device__ __noinline__ PackedRes wave_eval_compact(DevScene sc, DevOptions o, const uint16_t* myrow,
                                                    unsigned long long pending, PackedReq q, PackedRes out) {
    const int lane = lane_id();
    const int q_refsel = q.obj_mode ? 0 : q.ref;
    const unsigned long long rowbits = (unsigned long long)myrow;
    const int row_lo = (int)(rowbits & 0xffffffffull), row_hi = (int)(rowbits >> 32);
    const bool has_req = (pending >> lane) & 1ull;

    // ---------------- phase 1: gates only
    unsigned long long mymask = 0ull;  // requesting lanes: bit i <=> image i of the list passes the gates
    EV_T(e0);
    for (unsigned long long rem = pending; rem;) {
        EV_ACC(1, 0, 1);
        unsigned long long batch = 0ull;
        int J = 0;
        for (unsigned long long t = rem; t; t &= t - 1ull) {
            const int p = __ffsll((long long)t) - 1;
            const int np = bcast_i(q.nimg, p);
            if (batch && J + np > 64) break;
            batch |= 1ull << p;
            J += np;
        }
        rem &= ~batch;
        int myp = -1, myi = 0, myoff = 0;
        {
            int off = 0;
            for (unsigned long long t = batch; t; t &= t - 1ull) {
                const int p = __ffsll((long long)t) - 1;
                const int np = bcast_i(q.nimg, p);
                if (lane >= off && lane < off + np) { myp = p; myi = lane - off; }
                if (lane == p) myoff = off;
                off += np;
            }
        }
        const int src = myp >= 0 ? myp : 0;
        float c[3], zax[3], pxa[3], pya[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            c[k] = shfl_f(q.c[k], src); zax[k] = shfl_f(q.zax[k], src);
            pxa[k] = shfl_f(q.px[k], src); pya[k] = shfl_f(q.py[k], src);
        }
        const float scale = shfl_f(q.scale, src);
        const unsigned long long rb = ((unsigned long long)(unsigned int)shfl_i(row_hi, src) << 32) |
                                      (unsigned long long)(unsigned int)shfl_i(row_lo, src);
        int okj = 0;
        if (myp >= 0) okj = prep_image(sc.views[list_id<COHERENT>((const uint16_t*)rb, myi)], o, c, scale, pxa, pya, zax).ok;
        const unsigned long long okmask = __ballot(okj != 0);
        if ((batch >> lane) & 1ull) {
            const int np = q.nimg;
            const unsigned long long m = (np >= 64) ? ~0ull : ((1ull << np) - 1ull);
            mymask = (okmask >> myoff) & m;
        }
    }
    const bool refok = has_req && q.nimg > 0 && ((mymask >> q_refsel) & 1ull);
    if (has_req && !refok) {  // sampleTexture failed for the reference image (PatchOptimizer.cpp:291-293, 454-457)
        if (q.obj_mode) out.obj = 2.0;
        else {
            float s2 = 0.0f;
            for (int k = 0; k < q.nimg; k++) s2 += 2.0f;
            out.keep = 0ull;
            out.incc_sum = s2;
        }
    }
    const int mycnt = refok ? __popcll(mymask) : 0;  // grabs of this request, reference image included
    EV_T(e1);
    EV_ACC(0, e0, e1);

    // ---------------- phase 2: grabs
    for (unsigned long long rem = __ballot(refok); rem;) {
        EV_T(p0);
        unsigned long long batch = 0ull;
        int J = 0, maxnp = 0;
        for (unsigned long long t = rem; t; t &= t - 1ull) {
            const int p = __ffsll((long long)t) - 1;
            const int cp = bcast_i(mycnt, p);
            if (batch && J + cp > 64) break;
            batch |= 1ull << p;
            J += cp;
            const int np = bcast_i(q.nimg, p);
            maxnp = np > maxnp ? np : maxnp;
        }
        rem &= ~batch;
        // ---- lane -> (request myp, j-th usable image); every requesting lane learns where its grabs sit
        int myp = -1, myj = 0, myoff = 0;
        {
            int off = 0;
            for (unsigned long long t = batch; t; t &= t - 1ull) {
                const int p = __ffsll((long long)t) - 1;
                const int cp = bcast_i(mycnt, p);
                if (lane >= off && lane < off + cp) { myp = p; myj = lane - off; }
                if (lane == p) myoff = off;
                off += cp;
            }
        }
        const int src = myp >= 0 ? myp : 0;
        const unsigned long long pmask = ((unsigned long long)(unsigned int)shfl_i((int)(mymask >> 32), src) << 32) |
                                         (unsigned long long)(unsigned int)shfl_i((int)(mymask & 0xffffffffull), src);
        const int prefsel = shfl_i(q_refsel, src);
        const int poff = shfl_i(myoff, src);
        int myi = 0, refl = 0;
        if (myp >= 0) {
            unsigned long long m = pmask;
            for (int t = 0; t < myj; t++) m &= m - 1ull;  // drop the myj lowest usable images
            myi = __ffsll((long long)m) - 1;
            refl = poff + __popcll(pmask & ((1ull << prefsel) - 1ull));
        }
        float c[3], zax[3], pxa[3], pya[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            c[k] = shfl_f(q.c[k], src); zax[k] = shfl_f(q.zax[k], src);
            pxa[k] = shfl_f(q.px[k], src); pya[k] = shfl_f(q.py[k], src);
        }
        const float scale = shfl_f(q.scale, src);
        const unsigned long long rb = ((unsigned long long)(unsigned int)shfl_i(row_hi, src) << 32) |
                                      (unsigned long long)(unsigned int)shfl_i(row_lo, src);
        float t[3 * kTexPix];
#pragma unroll
        for (int k = 0; k < 3 * kTexPix; k++) t[k] = 0.0f;
        EV_T(p1);
        Prep P;
        P.ok = 0; P.tlx = P.tly = P.dxx = P.dxy = P.dyx = P.dyy = 0.0f; P.pix = nullptr; P.W = 0;
        if (myp >= 0) P = prep_image(sc.views[list_id<COHERENT>((const uint16_t*)rb, myi)], o, c, scale, pxa, pya, zax);
        EV_T(p2);
        if (myp >= 0) grab_window(P, t);   // P.ok holds: same inputs, same operations as in phase 1
        EV_T(p3);
        if (myp >= 0) normalize_window(t);
        EV_T(p4);
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 3 * kTexPix; k++) acc += shfl_f(t[k], refl) * t[k];
        const float raw = 1.0f - div_by(acc, 1.0 / 147.0);
        const float rob = robustincc(raw);
        EV_T(p5);

        // ---- delivery: every requesting lane of the pass walks its image list in index order
        const bool mine = (batch >> lane) & 1ull;
        const int np = mine ? q.nimg : 0;
        double val = 0.0;
        float isum = 0.0f;
        unsigned long long keep = 0ull;
        int nok = 0;
        for (int i = 0; i < maxnp; i++) {
            const bool usable = mine && ((mymask >> i) & 1ull);
            const int jl = (myoff + __popcll(mymask & ((1ull << i) - 1ull))) & 63;
            const float e_raw = shfl_f(raw, jl), e_rob = shfl_f(rob, jl);
            if (i < np) {
                if (q.obj_mode) {
                    if (usable && i != q_refsel) { val += (double)e_rob; nok++; }
                } else {
                    float e = 2.0f;
                    if (i == q_refsel) e = 0.0f;
                    else if (usable) e = q.robust ? e_rob : e_raw;
                    isum += e;
                    if (e < 1.0f - q.thr) keep |= 1ull << i;
                }
            }
        }
        if (mine) {
            out.ngrabs += mycnt;
            if (q.obj_mode) out.obj = (nok < o.MIN_IMAGES_PER_PATCH - 1) ? 2.0 : val / (double)nok;
            else { out.keep = keep; out.incc_sum = isum; }
        }
        EV_T(p6);
        EV_ACC(2, p0, p1); EV_ACC(3, p1, p2); EV_ACC(4, p2, p3); EV_ACC(5, p3, p4); EV_ACC(6, p4, p5); EV_ACC(7, p5, p6);
        EV_ACC(8, 0, 1); EV_ACC(9, 0, (long long)J);
    }
    return out;
}

}  // namespace hpmvs
