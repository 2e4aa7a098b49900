// The pruned render as three lean kernels instead of one large one ("deferred shadows").  Included by rt_device.hip only.
//
// What a path does - which rays it traces, which random draws it makes - does not depend on whether a shaded point sees a
// lamp: visibility only enters the colour.  So the path kernel traces the paths without tracing a single shadow ray.  For
// every ray it writes one node record (what was hit, the surface colour, N.L per lamp, and per lamp either the number of
// visible light samples where that is known without a ray, or zero and a job for the shadow kernel), linked to the
// path's previous node.  The shadow kernel then traces all jobs, 64 rays per wave with nothing else to keep in registers,
// and adds the visible counts into the node records.  The fold kernel walks every path's nodes backwards - the reverse of
// the reference's recursion order (CR/Scene.cu:23-58), so a node's children have been folded when the node is reached -
// and computes the colour with the arithmetic of CalnDiffusion / CalnReflection / CalnRefraction
// (CR/Scene_Material.cu:29-118), clamping at every level.  Same operations on the same operands as the one-kernel render:
// the image is bit-identical (tested), and so are the ray counts.
//
// Why: in the one-kernel render a shadow ray is traced with the whole path state alive (168 registers, ~70 spilled,
// 3 waves per SIMD).  Here the path kernel has no shadow code to make room for, and the shadow kernel holds a ray and
// nothing else, so both run with more waves per SIMD and without the spill traffic.
#ifndef RT_DEFER_HPP
#define RT_DEFER_HPP

#define RT_DEFER_LAMPS 4  // lamps a node record has room for; scenes with more render with the one-kernel path
struct alignas(16) NodeRec {  // 96 bytes
    uint32_t hdr;         // primitive index | ND_* flags
    uint32_t prev;        // previous node of the same path (RT_REF_END at the path's first node)
    double dist;          // hit distance (absorption on the way out of a refractive body)
    double color[3];      // terminal node: its colour; shaded node: material colour x texture
    double dt[RT_DEFER_LAMPS];    // N.L per lamp
    int32_t cnt[RT_DEFER_LAMPS];  // visible light samples per lamp
    uint32_t pad_[2];
};
static_assert(sizeof(NodeRec) == 96, "node record");
enum : uint32_t { ND_PRIM_MASK = 0x00FFFFFFu, ND_TERMINAL = 1u << 24, ND_NEED = 1u << 25, ND_REFL = 1u << 26, ND_REFR = 1u << 27, ND_FRONT = 1u << 28 };
struct JobRec {  // 64 bytes: the shadow rays of one shaded point towards one lamp
    double C[3];
    unsigned long long cand;
    uint32_t node, crashed, pixel, smp, draw, pad_[3];  // pad_[0] = tag of the launch that filled the slot
};
static_assert(sizeof(JobRec) == 64, "job record");
enum { DF_NODES = 0, DF_JOBS = 1, DF_SPANS = 1 + RT_DEFER_LAMPS, DF_WORDS = 1 + 2 * RT_DEFER_LAMPS };
struct DeferParams {
    NodeRec* nodes;
    JobRec* jobs;          // [lamp][jobs_cap]
    uint32_t* alloc;       // DF_WORDS counters: nodes handed out, jobs handed out per lamp, spans of jobs handed to the shadow kernel per lamp
    uint32_t* last_node;   // per path slot of the launch: its last node, RT_REF_END where the slot holds no path
    unsigned long long* overflow;  // set when a record did not fit: the host renders the call again with the one-kernel path
    uint32_t nodes_cap, jobs_cap;
    uint32_t launch_id;    // tag of this launch's job records (never 0): slots of the queue that hold another value are not in use
};

// Record slots are handed out to a wave a chunk at a time (one atomic per chunk instead of one per step of every wave on
// a single word); what is left of a chunk when it cannot hold a step's records is abandoned.  `cur` / `end` are the wave's
// chunk (the same value in every lane); returns the first of `n` consecutive slots.
#define RT_NODE_CHUNK 2048u
#define RT_JOB_CHUNK 1024u
DEV uint32_t take_slots(uint32_t* counter, uint32_t chunk, uint32_t n, uint32_t& cur, uint32_t& end) {
    if (cur + n > end) {
        uint32_t base = 0;
        if (lane_id() == (int)__ffsll((long long)__ballot(true)) - 1) base = atomicAdd(counter, chunk);
        cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        end = cur + chunk;
    }
    const uint32_t first = cur;
    cur += n;
    return first;
}

// ---- path kernel -----------------------------------------------------------------------------------------------------
// CalnDiffusion without the shadow rays: the surface colour, N.L per lamp and, per lamp, what is known about the shadow
// without tracing (light_candidates) go into the node record; where rays are needed, a job goes into that lamp's queue.
DEV void diffusion_defer(RPREF P, const DeferParams& D, const DevShade* shade, bool need, const Hit& h, uint32_t& draw, uint32_t pixel,
                         uint32_t smp, Counters& ctr, NodeRec* rec, uint32_t node, bool node_ok, uint32_t* job_chunk) {
    SCREF sc = P.sc;
    const int lane = lane_id();
    if (need) {
        const DevShade& m = shade[h.prim];
        d3 color = D3(m.color);
        if (m.texture >= 0) color = mulv(color, texture_color(sc, P.flags, m, h, ctr));
        ctr.shaded++;
        if (node_ok) rec->color[0] = color.x, rec->color[1] = color.y, rec->color[2] = color.z;
    }
    uint32_t off = 0;
    for (int l = 0; l < sc.n_lights; l++) {
        const rt_light RT_CONST& L = LIGHTS(sc)[l];
        const bool point = L.type == RT_LIGHT_POINT;
        const int n_l = point ? 1 : sc.shade_n;
        unsigned long long cand = ~0ull;
        bool work = need;
        int cnt = 0;
        double dt = 0.0;  // N.L of CR/Scene_Material.cu:50-51
        if (need) {
            ctr.rays_shadow += (uint32_t)n_l;
            dt = dot(unit(d3{L.O.x, L.O.y, L.O.z} - h.C), h.N);
            if (!(dt > RT_EPS)) {
                work = false;  // lamp behind the surface: the shade is not used (CR/Scene_Material.cu:52)
            } else {
                unsigned long long full, skip = 0ull;  // primitives the shadow test leaves out for this point and lamp
                if ((P.flags & RT_F_SHADOW_SELF_SKIP) && h.prim < 64) skip |= 1ull << h.prim;
                if (L.light_primitive >= 0 && L.light_primitive < 64) skip |= 1ull << L.light_primitive;
                const unsigned long long cell = candidate_cell(sc, h.C, l, skip, full);
                if (full != 0ull) cand = full;  // the whole cell lies in an umbra
                else cand = light_candidates(sc, P.flags, h.C, L, skip, cell, ~0ull, full) & ~skip;
                if (cand == 0ull) work = false, cnt = n_l;                      // nothing can stand in between
                else if ((full & cand) != 0ull) work = false, cnt = 0;          // in the umbra of one primitive
            }
            if (node_ok) rec->dt[l] = dt, rec->cnt[l] = cnt;  // cnt = 0 where rays decide: the shadow kernel adds the visible ones
        }
        const unsigned long long wmask = __ballot(work);
        if (wmask != 0ull) {
            const uint32_t n = (uint32_t)__popcll(wmask);
            uint32_t cur = job_chunk[2 * l], end = job_chunk[2 * l + 1];  // this wave's chunk of the lamp's queue (LDS)
            const uint32_t base = take_slots(D.alloc + DF_JOBS + l, RT_JOB_CHUNK, n, cur, end);
            if (lane == 0) job_chunk[2 * l] = cur, job_chunk[2 * l + 1] = end;
            ctr.traced_wave += n * (uint32_t)n_l;
            if (work) {
                const uint32_t j = base + (uint32_t)rank_in(wmask);
                if (j < D.jobs_cap && node_ok) {
                    JobRec& J = D.jobs[(size_t)l * D.jobs_cap + j];
                    J.pad_[0] = D.launch_id;  // marks the slot as filled by this launch (slots at the end of a chunk may stay unused)
                    J.C[0] = h.C.x, J.C[1] = h.C.y, J.C[2] = h.C.z;
                    J.cand = cand;
                    J.node = node, J.crashed = (uint32_t)h.prim, J.pixel = pixel, J.smp = smp, J.draw = draw + off;
                } else {
                    *D.overflow = 1ull;
                }
            }
        }
        if (!point) off += 2u * (uint32_t)n_l;
    }
    if (need) draw += off;
}

// All paths of a launch, as trace_stream traces them, minus everything that needs a colour.
DEV void trace_defer(RPREF P, const DeferParams& D, const DevShade* shade, Counters& ctr, uint32_t* stk, uint32_t* job_chunk) {
    SCREF sc = P.sc;
    // frame k = the hit at depth k+1 whose refraction waits for the reflection subtree (both coefficients non-zero)
    double f_C[RT_MAX_DEPTH][3], f_N[RT_MAX_DEPTH][3], f_V[RT_MAX_DEPTH][3];
    uint32_t f_prim[RT_MAX_DEPTH];  // primitive index | stage bits << 28
    uint32_t top = 0;
    int sp = 0;
    uint32_t my_path = 0, cur_node = RT_REF_END;
    bool tracing = false, drained = false;
    uint32_t range = blockIdx.x & 7u, ranges_done = 0;
    uint32_t pixel = 0, smp = 0, draw = 0, no_log = 0;
    d3 o = D3(0, 0, 0), d = D3(0, 0, 1);
    const uint32_t block_paths = 64u * (uint32_t)P.spp;
    const int lane = lane_id();
    uint32_t node_cur = 0, node_end = 0;  // this wave's chunk of node slots

    for (;;) {
        // ---- idle lanes start the next paths of the launch (CR/RayTracer.cu:52-72) ----
        const unsigned long long idle = __ballot(!tracing);
        if (!drained && (__popcll(idle) >= RT_REFILL_MIN || idle == ~0ull)) {
            const uint32_t n_new = (uint32_t)__popcll(idle);
            uint32_t base = 0, end = 0;
            for (;;) {
                const uint32_t first = range * P.range_paths;
                end = first + P.range_paths < P.n_paths ? first + P.range_paths : P.n_paths;
                if (first < end) {
                    if (lane == 0) base = atomicAdd(P.queue + range, n_new);
                    base = first + (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (base < end) break;
                }
                range = (range + 1u) & 7u;
                if (++ranges_done == 8) {
                    drained = true;
                    base = end = 0;
                    break;
                }
            }
            const uint32_t g = base + (uint32_t)rank_in(idle);
            if (!tracing && g < end) {
                const uint32_t blk = div_by(g, block_paths, P.inv_block_paths), r = g - blk * block_paths;
                const uint32_t slot = div_by(r, (uint32_t)P.spp, P.inv_spp), s = r - slot * (uint32_t)P.spp;
                const BlockRec rec = P.blocks[blk];
                if ((slot & 7u) < rec.w && (slot >> 3) < rec.h) {
                    const int px = rec.x0 + (int)(slot & 7u), py = rec.y0 + (int)(slot >> 3);
                    int x, y;
                    map_pixel(P.flags, P.tex_w, P.tex_h, sc.cam_w, sc.cam_h, px, py, x, y);
                    const d3 dc = unit(emit(sc, x, y));
                    pixel = (uint32_t)py * (uint32_t)P.tex_w + (uint32_t)px;
                    smp = s;
                    my_path = (blk * (uint32_t)P.spp + s) * 64u + slot;
                    sp = 0, draw = 0, tracing = true, cur_node = RT_REF_END;
                    o = D3(sc.camO), d = dc;
                    if (P.spp == 1) {
                        if (!(P.flags & RT_F_NORMALIZE_PRIMARY)) d = emit(sc, x, y);
                    } else {
                        float fu, fv;
                        frand_pair(P.k0, P.k1, pixel, smp, 0u, fu, fv);
                        draw = 2;
                        double u = x + (2.0f * fu - 1.0f);
                        double v = y + (2.0f * fv - 1.0f);
                        d = emit(sc, u, v);
                        if (P.flags & RT_F_NORMALIZE_PRIMARY) d = unit(d);
                    }
                    if (P.dof) {  // modifyRayForDepthOfField, CR/RayTracer.cu:31-38
                        const d3 focal = D3(sc.camO) + dc * P.focal;
                        float fa, fb;
                        frand_pair(P.k0, P.k1, pixel, smp, draw, fa, fb);
                        draw += 2;
                        double rx = (fa - 0.5) * P.aperture / 10.0;
                        double ry = (fb - 0.5) * P.aperture / 10.0;
                        o = o + (D3(sc.camDx) * rx + D3(sc.camDy) * ry);
                        d = unit(focal - o);
                    }
                }
            }
        }
        const unsigned long long tm = __ballot(tracing);
        if (tm == 0ull) {
            if (drained) break;
            continue;  // every slot just handed out lies outside its tile
        }
        // ---- one node per ray ----
        const uint32_t nbase = take_slots(D.alloc + DF_NODES, RT_NODE_CHUNK, (uint32_t)__popcll(tm), node_cur, node_end);
        const uint32_t my_node = nbase + (uint32_t)rank_in(tm);
        const bool node_ok = tracing && my_node < D.nodes_cap;
        if (tracing && !node_ok) *D.overflow = 1ull;
        NodeRec* rec = D.nodes + (node_ok ? my_node : 0u);

        Hit h;
        h.prim = -1, h.leaf = -1, h.dist = RT_BIG_DIST, h.front = false, h.N = D3(0, 0, 0), h.C = D3(0, 0, 0);
        bool surface = false;
        d3 result = D3(0, 0, 0);  // of a terminal node: sky, lamp, or black beyond the depth limit (CR/Scene.cu:25)
        if (tracing && sp + 1 <= RT_MAX_DEPTH) {
            if (sp > 0) ctr.rays_secondary++;
            h = scene_closest<false, false>(sc, P.flags, o, d, ctr, nullptr, 0, no_log, stk, nullptr);
            if (h.prim < 0) {
                d3 c = unit(d);
                double t = 0.5 * (((P.flags & RT_F_SKY_Y) ? c.y : c.z) + 1.0);
                result = clamp1(D3(sc.bg_bottom) * (1.0 - t) + D3(sc.bg_top) * t);
            } else if (shade[h.prim].is_light) {
                result = clamp1(D3(shade[h.prim].color));
            } else {
                surface = true;
            }
        }
        bool need_shade = false, has_refl = false, has_refr = false;
        if (surface) {
            const DevShade& m = shade[h.prim];
            need_shade = m.diff > RT_EPS || m.spec > RT_EPS;
            has_refl = m.refl > RT_EPS, has_refr = m.refr > RT_EPS;
        }
        if (node_ok) {
            rec->prev = cur_node;
            if (!surface) {
                rec->hdr = ND_TERMINAL;
                rec->color[0] = result.x, rec->color[1] = result.y, rec->color[2] = result.z;
            } else {
                rec->hdr = (uint32_t)h.prim | (need_shade ? ND_NEED : 0u) | (has_refl ? ND_REFL : 0u) | (has_refr ? ND_REFR : 0u) | (h.front ? ND_FRONT : 0u);
                rec->dist = h.dist;
            }
        }
        if (tracing) cur_node = my_node;
        if (__ballot(need_shade) != 0ull) diffusion_defer(P, D, shade, need_shade, h, draw, pixel, smp, ctr, rec, my_node, node_ok, job_chunk);

        if (tracing) {
            bool have_result = !surface;
            if (surface) {
                const DevShade& m = shade[h.prim];
                if (has_refl || has_refr) {
                    const int k = sp;
                    if (has_refl && has_refr) {  // the refraction waits for the reflection subtree and needs the hit again
                        f_C[k][0] = h.C.x, f_C[k][1] = h.C.y, f_C[k][2] = h.C.z;
                        f_N[k][0] = h.N.x, f_N[k][1] = h.N.y, f_N[k][2] = h.N.z;
                        f_V[k][0] = d.x, f_V[k][1] = d.y, f_V[k][2] = d.z;
                    }
                    uint32_t st = h.front ? ST_FRONT : 0;
                    d3 nd;
                    if (has_refl) {  // CalnReflection, CR/Scene_Material.cu:68-99
                        st |= ST_REFL | (has_refr ? ST_PENDING_REFR : 0);
                        d3 rv = reflect(d, h.N);
                        const int dep = sp + 1;
                        if (m.drefl < RT_EPS || dep > RT_MAX_DREFL_DEP) {
                            nd = rv;
                        } else {
                            d3 bx = vertical(rv);
                            d3 by = unit(cross(rv, bx));
                            bx = bx * m.drefl;
                            by = by * m.drefl;
                            float fx, fy;  // GetBlur, CR/Scene_Material.cu:19-27
                            frand_pair(P.k0, P.k1, pixel, smp, draw, fx, fy);
                            draw += 2;
                            double x = fx, y = fy;
                            double sn, cs;
                            det_sincos(y, sn, cs);
                            double bxk = x * cs, byk = x * sn;
                            nd = rv + bx * bxk + by * byk;
                        }
                    } else {  // CalnRefraction, CR/Scene_Material.cu:101-118
                        st |= ST_REFR;
                        double n = m.rindex;
                        if (h.front) n = 1.0 / n;
                        nd = refract(d, h.N, n);
                    }
                    top = (uint32_t)h.prim | st << 28;
                    f_prim[k] = top;
                    sp++;
                    o = h.C;
                    d = nd;
                } else {
                    have_result = true;
                }
            }
            // a finished subtree: back to the frame that waits for it
            while (have_result) {
                if (sp == 0) {  // the path is complete
                    D.last_node[my_path] = cur_node;
                    tracing = false;
                    break;
                }
                const int k = sp - 1;
                const uint32_t fp = top, st = fp >> 28;
                if ((st & ST_REFL) && (st & ST_PENDING_REFR)) {
                    const DevShade& m = shade[fp & 0x0fffffffu];
                    top = (fp & 0x0fffffffu) | ((st & ST_FRONT) | ST_REFR) << 28;
                    f_prim[k] = top;
                    double n = m.rindex;
                    if (st & ST_FRONT) n = 1.0 / n;
                    d3 V = D3(f_V[k][0], f_V[k][1], f_V[k][2]), N = D3(f_N[k][0], f_N[k][1], f_N[k][2]);
                    o = D3(f_C[k][0], f_C[k][1], f_C[k][2]);
                    d = refract(V, N, n);
                    have_result = false;  // the refraction child is traced next
                } else {
                    sp--;
                    if (sp > 0) top = f_prim[sp - 1];
                }
            }
        }
    }
}

#ifndef RT_PATH_WAVES
#define RT_PATH_WAVES 3
#endif
#ifndef RT_SHADOW_WAVES
#define RT_SHADOW_WAVES 4
#endif

DEV void flush_counters(RPREF P, Counters& ctr) {
    const int lane = lane_id();
    unsigned long long v[CTR_COUNT] = {ctr.rays_secondary, ctr.rays_shadow, ctr.nodes, ctr.tris, ctr.shaded, ctr.texfetch, 0ull};
#pragma unroll
    for (int i = 0; i < CTR_COUNT; i++) {
        unsigned long long s = wave_sum_u64(v[i]);
        if (lane == 0 && s) atomicAdd(&P.counters[i], s);
    }
    if (lane == 0 && ctr.traced_wave) atomicAdd(&P.counters[CTR_TRACED], (unsigned long long)ctr.traced_wave);
}

template <int SHADE_CAP>
__global__ __launch_bounds__(256, RT_PATH_WAVES) void path_kernel(RenderParams, DeferParams D) {
    RPREF P = KERNARG(RenderParams);
    const int lane = lane_id();
    __shared__ uint32_t s_stack[4 * RT_STACK * 64];
    uint32_t* stk = s_stack + (threadIdx.x >> 6) * (RT_STACK * 64) + lane;
    __shared__ DevShade s_shade[SHADE_CAP > 0 ? SHADE_CAP : 1];
    const DevShade* shade = P.sc.shade;
    if (SHADE_CAP > 0) {
        const int words = P.sc.n_prims * (int)(sizeof(DevShade) / 4);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(P.sc.shade);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_shade);
        for (int i = threadIdx.x; i < words; i += 256) dst[i] = src[i];
        __syncthreads();
        shade = s_shade;
    }
    Counters ctr;
    ctr.rays_secondary = ctr.rays_shadow = ctr.nodes = ctr.tris = ctr.shaded = ctr.texfetch = 0;
    ctr.traced_wave = 0;
    __shared__ uint32_t s_job_chunk[4][2 * RT_DEFER_LAMPS];  // per wave and lamp: its chunk of the job queue (next slot, end)
    uint32_t* job_chunk = s_job_chunk[threadIdx.x >> 6];
    if (lane < 2 * RT_DEFER_LAMPS) job_chunk[lane] = 0;
    trace_defer(P, D, shade, ctr, stk, job_chunk);
    flush_counters(P, ctr);
}

// ---- shadow kernel ---------------------------------------------------------------------------------------------------
// Every job of every lamp: lane = one light sample of one shaded point.  The rays of a job sit in neighbouring lanes; the
// first lane of each run adds the run's visible samples to the node's count (integer adds: the order does not matter).
template <bool SHARE>
__global__ __launch_bounds__(256, RT_SHADOW_WAVES) void shadow_kernel(RenderParams, DeferParams D) {
    RPREF P = KERNARG(RenderParams);
    if (*D.overflow) return;
    SCREF sc = P.sc;
    const int lane = lane_id();
    __shared__ uint32_t s_stack[4 * RT_STACK * 64];
    uint32_t* stk = s_stack + (threadIdx.x >> 6) * (RT_STACK * 64) + lane;
    __shared__ __attribute__((aligned(8))) uint32_t s_xch[SHARE ? 4 * 192 : 2];
    uint32_t* xch = SHARE ? s_xch + (threadIdx.x >> 6) * 192 : nullptr;
    Counters ctr;
    ctr.rays_secondary = ctr.rays_shadow = ctr.nodes = ctr.tris = ctr.shaded = ctr.texfetch = 0;
    ctr.traced_wave = 0;
    for (int l = 0; l < sc.n_lights && l < RT_DEFER_LAMPS; l++) {
        const rt_light RT_CONST& L = LIGHTS(sc)[l];
        const bool point = L.type == RT_LIGHT_POINT;
        const uint32_t n_l = point ? 1u : (uint32_t)sc.shade_n;
        uint32_t n_jobs = D.alloc[DF_JOBS + l];
        if (n_jobs > D.jobs_cap) n_jobs = D.jobs_cap;
        const JobRec* jobs = D.jobs + (size_t)l * D.jobs_cap;
        // A wave takes the jobs a span at a time.  A span is one allocation chunk of the path kernel: the jobs one of its waves
        // wrote one after another, i.e. shaded points next to each other in the image, whose rays walk the same part of the
        // scene - the wave's node fetches then hit in its CU's cache, which rays dealt round-robin over the waves would not.
        for (;;) {
            uint32_t span = 0;
            if (lane == 0) span = atomicAdd(D.alloc + DF_SPANS + l, 1u);
            span = (uint32_t)__builtin_amdgcn_readfirstlane((int)span);
            const uint32_t j0 = span * RT_JOB_CHUNK;
            if (j0 >= n_jobs) break;
            const uint32_t j1 = j0 + RT_JOB_CHUNK < n_jobs ? j0 + RT_JOB_CHUNK : n_jobs;
            const uint32_t total = (j1 - j0) * n_l;
            for (uint32_t r0 = 0; r0 < total; r0 += 64u) {
                const uint32_t r = r0 + (uint32_t)lane;
                const bool in_range = r < total;
                const uint32_t job = j0 + (in_range ? r / n_l : 0u), si = in_range ? r % n_l : 0u;
                const JobRec& J = jobs[job];
                const bool act = in_range && J.pad_[0] == D.launch_id;  // slots at the end of a wave's chunk may be unused
                if (__ballot(act) == 0ull) continue;
                const d3 C = D3(J.C[0], J.C[1], J.C[2]);
                d3 target = d3{L.O.x, L.O.y, L.O.z};
                if (act && !point) {
                    float fa, fb;
                    frand_pair(P.k0, P.k1, J.pixel, J.smp, J.draw + 2u * si, fa, fb);
                    target = light_sample(L, fa, fb);
                }
                bool clear = false;
                const unsigned long long wave_prims = wave_or_u64(act ? J.cand : 0ull);
                if (SHARE) clear = shadow_clear_shared(sc, P.flags, act, C, target, J.cand, wave_prims, ctr, stk, xch);
                else if (act) clear = shadow_clear<false>(sc, P.flags, C, target, (int)J.crashed, L.light_primitive, J.cand, wave_prims, ctr, stk);
                // runs of lanes that belong to one job
                const uint32_t job_before = (uint32_t)__shfl_up((int)job, 1, 64);
                const bool leader = act && (lane == 0 || job_before != job);
                const unsigned long long leaders = __ballot(leader), cm = __ballot(clear && act), am = __ballot(act);
                if (leader) {
                    const unsigned long long after = lane == 63 ? 0ull : (leaders >> (lane + 1)) << (lane + 1);  // leaders above this lane
                    const int end = after ? (int)__ffsll((long long)after) - 1 : 64;                          // first lane of the next run
                    unsigned long long run = (end >= 64 ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull) & am;
                    const int c = __popcll(cm & run);
                    if (c) atomicAdd(&D.nodes[J.node].cnt[l], c);
                }
            }
        }
    }
    flush_counters(P, ctr);
}

// ---- fold kernel -----------------------------------------------------------------------------------------------------
// lane = path.  Walks the path's nodes from the last to the first; a node's children (reflection subtree first, then
// refraction subtree, in the recursion's order) lie after it, so walking backwards meets the refraction subtree's root
// first and the reflection subtree's root last: their colours wait on a small stack, reflection on top.
__global__ __launch_bounds__(256) void fold_kernel(RenderParams, DeferParams D) {
    RPREF P = KERNARG(RenderParams);
    if (*D.overflow) return;
    SCREF sc = P.sc;
    for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < P.n_paths; p += gridDim.x * 256u) {
        uint32_t idx = D.last_node[p];
        if (idx == RT_REF_END) continue;
        double stack[RT_MAX_DEPTH + 4][3];
        int sp = 0;
        while (idx != RT_REF_END) {
            const NodeRec& r = D.nodes[idx];
            const uint32_t hdr = r.hdr;
            d3 out;
            if (hdr & ND_TERMINAL) {
                out = D3(r.color[0], r.color[1], r.color[2]);
            } else {
                const DevShade& m = sc.shade[hdr & ND_PRIM_MASK];
                d3 acc = D3(0, 0, 0);
                if (hdr & ND_NEED) {  // CalnDiffusion, CR/Scene_Material.cu:29-66
                    const d3 color = D3(r.color[0], r.color[1], r.color[2]);
                    const double m_diff = m.diff, m_spec = m.spec;
                    d3 ret = D3(0, 0, 0);
                    for (int l = 0; l < sc.n_lights; l++) {
                        const rt_light RT_CONST& L = LIGHTS(sc)[l];
                        const bool point = L.type == RT_LIGHT_POINT;
                        const int n_l = point ? 1 : sc.shade_n;
                        const double dt = r.dt[l];
                        const int cnt = r.cnt[l];
                        if (dt > RT_EPS) {
                            double shade = point ? (cnt ? 1.0 : 0.0) : (double)cnt / n_l;
                            if (!(shade < RT_EPS)) {
                                const d3 Lc = d3{L.color.x, L.color.y, L.color.z};
                                if (m_diff > RT_EPS) {
                                    double diff = m_diff * dt * shade;
                                    ret = ret + mulv(color, Lc) * diff;
                                }
                                if (m_spec > RT_EPS) {
                                    const double p2 = dt * dt, p4 = p2 * p2, p5 = p4 * dt, p10 = p5 * p5;
                                    double spec = m_spec * (p10 * p10) * shade;
                                    ret = ret + mulv(color, Lc) * spec;
                                }
                            }
                        }
                    }
                    acc = acc + ret;
                }
                if (hdr & ND_REFL) {  // CalnReflection's weight, CR/Scene_Material.cu:98
                    sp = sp > 0 ? sp - 1 : 0;
                    const d3 res = D3(stack[sp][0], stack[sp][1], stack[sp][2]);
                    acc = acc + mulv(res, D3(m.color)) * m.refl;
                }
                if (hdr & ND_REFR) {  // CalnRefraction's weight and absorption, CR/Scene_Material.cu:112-117
                    sp = sp > 0 ? sp - 1 : 0;
                    const d3 res = D3(stack[sp][0], stack[sp][1], stack[sp][2]);
                    d3 contrib;
                    if (hdr & ND_FRONT) {
                        contrib = res * m.refr;
                    } else {
                        d3 ab = D3(m.absor) * -r.dist;
                        d3 tr = D3(rt_exp(ab.x), rt_exp(ab.y), rt_exp(ab.z));
                        contrib = mulv(res, tr) * m.refr;
                    }
                    acc = acc + contrib;
                }
                out = clamp1(acc);
            }
            if (sp < RT_MAX_DEPTH + 4) stack[sp][0] = out.x, stack[sp][1] = out.y, stack[sp][2] = out.z, sp++;
            idx = r.prev;
        }
        double* dst = P.path_out + (size_t)p * 3;
        dst[0] = stack[0][0], dst[1] = stack[0][1], dst[2] = stack[0][2];
    }
}
#endif
