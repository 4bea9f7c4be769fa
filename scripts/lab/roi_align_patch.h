// LAB (not built into the library): the r03 'patch through LDS' fp16 ROIAlign forward.  Correct - bit-identical to the shipped
// separable kernel on every test - but SLOWER: 1.55 vs 1.21 ms on scripts/roi_time.py's synthetic load, 1.24 vs 0.89 ms per launch
// in the network (bench.py: 880 vs 896 pairs/s).  The shipped kernel's per-bin loads already hit L1 / L2; staging the patch costs
// two barriers per 32-channel chunk, 48 KiB of LDS per workgroup (3 workgroups per CU instead of 8) and 196 of 256 threads busy.
// Was spliced into csrc/roi_align.hip before the backward section; pe_roi_align_nhwc chose it for dtype fp16 when C % 32 == 0.
// ---- fp16 forward, third generation (r03): the ROI's feature patch goes through LDS ONCE per channel chunk -------------------
// The separable kernel above still fetches every pixel of a bin per (bin, channel vector): neighbouring bins share their
// border cells and the 32 channel vectors of a bin walk the same cells, so an ROI pulls ~0.5 MB through L1 / L2 for a 0.2 MB
// patch, in 16-byte pieces that are 512 B apart (rocprof r02: 2.55 GB of HBM traffic per launch against 1.9 GB algorithmic, at
// 2.9 TB/s).  Here a workgroup copies the patch rows [ymin, ymax) x [xmin, xmax) of CH = 32 channels (64 contiguous bytes per
// cell = one fetch granule) into LDS with up to 12 independent 16-byte loads per thread, then the 7 x 7 bins x 4 channel
// vectors read their pixels from LDS.  Same tables, same pixel order, same fused multiply-adds as the separable kernel: the
// results are bit-identical to it.  ROIs whose patch exceeds PATCH_MAX_CELLS (or that need the tap fallback) run the old loop.
constexpr int PATCH_CH = 32;             // channels per chunk
constexpr int PATCH_MAX_CELLS = 768;     // 48 KiB of LDS per workgroup -> 3 workgroups per CU
constexpr int PATCH_THREADS = 256;
constexpr int PATCH_PIECES = PATCH_MAX_CELLS * (PATCH_CH / 8) / PATCH_THREADS;   // 16-byte pieces per thread and chunk: 12

__global__ __launch_bounds__(PATCH_THREADS) void roi_align_patch_kernel(RoiArgs a) {
    using T = _Float16;
    constexpr int V = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char patch[];
    const int r = blockIdx.x;
    int b;
    float bx1, by1, bx2, by2;
    bool live = true;
    if (a.rois5) {
        const float* p = a.rois + (size_t)r * 5;
        b = (int)p[0]; bx1 = p[1]; by1 = p[2]; bx2 = p[3]; by2 = p[4];
    } else {
        b = r / a.per_image;
        const float* p = a.rois + (size_t)r * 4;
        bx1 = p[0]; by1 = p[1]; bx2 = p[2]; by2 = p[3];
        if (a.counts && (r - b * a.per_image) >= a.counts[b]) live = false;
    }
    int lvl = 0;
    if (a.num_levels > 1) {
        const float area = (bx2 - bx1) * (by2 - by1);
        const float sz = sqrtf(area);
        float lv = floorf((float)a.canonical_level + log2f(sz / a.canonical_size + 2.220446049250313e-16f));
        lv = fminf(fmaxf(lv, (float)a.min_level), (float)a.max_level);
        lvl = (int)lv - a.min_level;
    }
    if (a.out_level && threadIdx.x == 0) a.out_level[r] = live ? lvl : -1;
    const int H = a.fh[lvl], W = a.fw[lvl];
    const float scale = a.scale[lvl];
    const T* feat = reinterpret_cast<const T*>(a.feat[lvl]) + (size_t)b * H * W * a.C;
    const float offset = a.aligned ? 0.5f : 0.0f;
    const float start_w = bx1 * scale - offset, start_h = by1 * scale - offset;
    const float end_w = bx2 * scale - offset, end_h = by2 * scale - offset;
    float roi_w = end_w - start_w, roi_h = end_h - start_h;
    if (!a.aligned) { roi_w = fmaxf(roi_w, 1.f); roi_h = fmaxf(roi_h, 1.f); }
    const float bin_h = roi_h / (float)a.ph, bin_w = roi_w / (float)a.pw;
    const int grid_h = a.sampling_ratio > 0 ? a.sampling_ratio : (int)ceilf(roi_h / (float)a.ph);
    const int grid_w = a.sampling_ratio > 0 ? a.sampling_ratio : (int)ceilf(roi_w / (float)a.pw);
    const float count = (float)max(grid_h * grid_w, 1);
    T* out = reinterpret_cast<T*>(a.out) + (size_t)r * a.ph * a.pw * a.C;
    const int tid = threadIdx.x;
    const int cvec = a.C / V;
    const int items = a.ph * a.pw * cvec;
    if (!live) {      // padded slot: zeros (block-uniform)
        const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int it = tid; it < items; it += PATCH_THREADS) *reinterpret_cast<half8*>(out + (size_t)it * V) = zero;
        return;
    }
    __shared__ SepTables tabs;
    __shared__ int ext[4];     // ymin, ymax (exclusive), xmin, xmax
    if (tid == 0) tabs.fallback = 0;
    __syncthreads();
    if (tid < a.ph) sep_build_axis(tabs, 0, tid, start_h + tid * bin_h, bin_h, grid_h, H);
    else if (tid < a.ph + a.pw) sep_build_axis(tabs, 1, tid - a.ph, start_w + (tid - a.ph) * bin_w, bin_w, grid_w, W);
    __syncthreads();
    if (tid == 0) {
        int lo[2] = {0x7fffffff, 0x7fffffff}, hi[2] = {0, 0};
        for (int ax = 0; ax < 2; ++ax)
            for (int k = 0; k < (ax ? a.pw : a.ph); ++k)
                if (tabs.n[ax][k] > 0) { lo[ax] = min(lo[ax], tabs.first[ax][k]); hi[ax] = max(hi[ax], tabs.first[ax][k] + tabs.n[ax][k]); }
        ext[0] = lo[0]; ext[1] = hi[0]; ext[2] = lo[1]; ext[3] = hi[1];
    }
    __syncthreads();
    const int ymin = ext[0], xmin = ext[2];
    const int PH = max(ext[1] - ext[0], 0), PW = max(ext[3] - ext[2], 0);
    const int cells = PH * PW;
    if (tabs.fallback || cells > PATCH_MAX_CELLS) {
        // ---- large / degenerate ROI: the second-generation loops, unchanged ----
        for (int it = tid; it < items; it += PATCH_THREADS) {
            const int cv = it % cvec, bin = it / cvec;
            const int ph = bin / a.pw, pw = bin - ph * a.pw;
            float acc[V];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = 0.f;
            if (!tabs.fallback) {
                const int ny = tabs.n[0][ph], nx = tabs.n[1][pw];
                const float* wyr = tabs.w[0][ph];
                const float* wxr = tabs.w[1][pw];
                for (int rr = 0; rr < ny; ++rr)
                    for (int c = 0; c < nx; ++c) {
                        const half8 h = *reinterpret_cast<const half8*>(feat + ((size_t)(tabs.first[0][ph] + rr) * W + tabs.first[1][pw] + c) * a.C + cv * V);
                        const float wv = wyr[rr] * wxr[c];
#pragma unroll
                        for (int e = 0; e < V; ++e) acc[e] = __builtin_fmaf((float)h[e], wv, acc[e]);
                    }
            } else {
                for (int iy = 0; iy < grid_h; ++iy) {
                    const float yy = start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)grid_h;
                    for (int ix = 0; ix < grid_w; ++ix) {
                        const float xx = start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)grid_w;
                        float x = xx, y = yy;
                        if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
                        if (y <= 0) y = 0;
                        if (x <= 0) x = 0;
                        int y_low = (int)y, x_low = (int)x, y_high, x_high;
                        if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
                        if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
                        const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
                        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                        float v1[V], v2[V], v3[V], v4[V];
                        Vec<T>::load(feat + ((size_t)y_low * W + x_low) * a.C + cv * V, v1);
                        Vec<T>::load(feat + ((size_t)y_low * W + x_high) * a.C + cv * V, v2);
                        Vec<T>::load(feat + ((size_t)y_high * W + x_low) * a.C + cv * V, v3);
                        Vec<T>::load(feat + ((size_t)y_high * W + x_high) * a.C + cv * V, v4);
#pragma unroll
                        for (int e = 0; e < V; ++e) acc[e] += w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] /= count;
            Vec<T>::store(out + (size_t)bin * a.C + cv * V, acc);
        }
        return;
    }
    // ---- patch route ----
    constexpr int Q = PATCH_CH / 8;                       // 16-byte pieces per cell and chunk
    int g_off[PATCH_PIECES];                              // element offset of this thread's piece inside the image level, or -1
#pragma unroll
    for (int k = 0; k < PATCH_PIECES; ++k) {
        const int idx = tid + k * PATCH_THREADS;
        const int cell = idx / Q, q = idx - cell * Q;
        const int rr = cell / max(PW, 1), c = cell - rr * PW;
        g_off[k] = cell < cells ? ((ymin + rr) * W + xmin + c) * a.C + q * 8 : -1;
    }
    // this thread's bin / channel vector in the compute phase: item = bin * Q + v  (ph * pw * Q <= 256 items)
    const int nitem = a.ph * a.pw * Q;
    const int bin = tid / Q, v = tid - bin * Q;
    const int ph = bin / a.pw, pw = bin - ph * a.pw;
    const bool worker = tid < nitem;
    const int ny = worker ? tabs.n[0][ph] : 0, nx = worker ? tabs.n[1][pw] : 0;
    const int l_base = worker ? (((tabs.first[0][ph] - ymin) * PW + (tabs.first[1][pw] - xmin)) * Q + v) * 16 : 0;
    const float* wyr = tabs.w[0][worker ? ph : 0];
    const float* wxr = tabs.w[1][worker ? pw : 0];
    half8 stage[PATCH_PIECES];
#pragma unroll
    for (int k = 0; k < PATCH_PIECES; ++k)
        if (g_off[k] >= 0) stage[k] = *reinterpret_cast<const half8*>(feat + g_off[k]);
    for (int ch0 = 0; ch0 < a.C; ch0 += PATCH_CH) {
        if (ch0) __syncthreads();                         // the previous chunk's readers are done
#pragma unroll
        for (int k = 0; k < PATCH_PIECES; ++k)
            if (g_off[k] >= 0) *reinterpret_cast<half8*>(patch + (size_t)(tid + k * PATCH_THREADS) * 16) = stage[k];
        if (ch0 + PATCH_CH < a.C) {                       // the next chunk's pieces fly under this chunk's bins
#pragma unroll
            for (int k = 0; k < PATCH_PIECES; ++k)
                if (g_off[k] >= 0) stage[k] = *reinterpret_cast<const half8*>(feat + g_off[k] + ch0 + PATCH_CH);
        }
        __syncthreads();
        if (worker) {
            float acc[V];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = 0.f;
            // pixels in row-major order, four LDS reads in flight; (rr, c) and the byte offset advance by selects and adds
            const int npx = ny * nx;
            const int step_c = Q * 16, step_r = (PW - nx + 1) * Q * 16;
            int rr = 0, c = 0, poff = l_base;
            for (int i = 0; i < npx; i += 4) {
                half8 h[4];
                float w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = i + u < npx;
                    const float wv = wyr[rr] * wxr[c];
                    w[u] = ok ? wv : 0.f;
                    h[u] = *reinterpret_cast<const half8*>(patch + poff);
                    const bool adv = i + u + 1 < npx;
                    const bool wrap = adv && c + 1 == nx;
                    poff += wrap ? step_r : (adv ? step_c : 0);
                    c = wrap ? 0 : c + (adv ? 1 : 0);
                    rr += wrap ? 1 : 0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] = __builtin_fmaf((float)h[u][e], w[u], acc[e]);
            }
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] /= count;
            Vec<T>::store(out + (size_t)bin * a.C + ch0 + v * V, acc);
        }
    }
}

