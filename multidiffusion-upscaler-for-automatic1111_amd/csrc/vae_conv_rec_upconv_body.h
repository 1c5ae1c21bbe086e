// Body of k_upconv_rec / k_upconv_rec_st (csrc/vae_conv_rec.hip includes this file twice; see vae_conv_rec_direct_body.h for why the text is
// shared by inclusion and not through a body template): MDT_REC_ST = 1 adds the GroupNorm statistics of the output.
__global__ __launch_bounds__(512, 2) void MDT_REC_KERNEL(const ConvRParams P) {
    constexpr bool ST = MDT_REC_ST != 0;
    constexpr int MT = 4, MW = 2, WM = 2, NROW = 2, TH = 8;
    constexpr int ROWS = TH + 2, COLS = 34;
    using IS = InStage<ROWS>;
    constexpr int W_REC = 2 * 2 * 2 * MT * 64, W_DMA = W_REC / 64, W_PW = W_DMA / 8;
    __shared__ u32x4 smem[2 * IS::PAD + 3 * W_REC + 2 * EC_REC];
    u32x4* const in_l = smem;
    u32x4* const w_l = smem + 2 * IS::PAD;
    u32x4* const ec_l = smem + 2 * IS::PAD + 3 * W_REC;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wr = wave / WM;
    const int Hp = P.HinF + 2, Wp = rec_pitch(P.WinF), Pn = P.Cin >> 3;      // pitches of the WHOLE input image; items tile its window
    const size_t plane = (size_t)Hp * Wp;

    struct Item {
        int b, cb, a, y0, x0;   // y0, x0: INPUT coordinates (relative to the window)
    };
    const int per = P.NCB * 2, per_img = ((P.ptiles + 7) / 8) * 8 * per, total = per_img * P.B;
    auto decode = [&](int work, Item& it) -> bool {
        it.b = work / per_img;
        const int r = work - it.b * per_img, xcd = r & 7, slot = r >> 3;
        const int ptile = (slot / per) * 8 + xcd, rem = slot % per;
        it.cb = rem >> 1;
        it.a = rem & 1;
        const int py = ptile / P.PX, px = ptile - py * P.PX;
        it.y0 = py * TH;
        it.x0 = px * 32;
        return ptile < P.ptiles;
    };
    auto next_valid = [&](int work, Item& it) -> int {
        while (work < total && !decode(work, it)) work += gridDim.x;
        return work;
    };
    auto make_ioff = [&](const Item& it, unsigned (&ioff)[IS::PW]) {
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) {
            const int di = wave + 8 * i;
            int s = (di % IS::HALF_DMA) * 64 + lane;
            if (s >= IS::HALF) s = IS::HALF - 1;
            const int g = s / (ROWS * COLS), p = s - g * (ROWS * COLS);
            const int r = p / COLS, c = p - r * COLS;
            int pr = P.iy0[it.b & (REC_WIN_MAXB - 1)] + it.y0 + r, pc = P.ix0[it.b & (REC_WIN_MAXB - 1)] + it.x0 + c;     // inside the window's own border: the image's real neighbours
            pr = pr < Hp ? pr : Hp - 1;
            pc = (pc < P.WinF + 1 ? pc : P.WinF + 1) + REC_COL0;
            ioff[i] = (unsigned)(((size_t)g * plane + (size_t)pr * Wp + pc) * 16);
        }
    };
    auto issue_input = [&](const Item& it, const unsigned (&ioff)[IS::PW], int k, int stage) {
        const char* xb = reinterpret_cast<const char*>(P.x + (size_t)it.b * 2 * Pn * plane);
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) {
            const int di = wave + 8 * i;
            if (di < IS::DMA) {
                const char* base = xb + ((size_t)(di / IS::HALF_DMA) * Pn + 2 * (size_t)k) * plane * 16;
                dma16(base, ioff[i], in_l + stage * IS::PAD + di * 64);
            }
        }
    };
    const int nph = P.NK * 2;
    const unsigned lane16 = lane * 16;
    auto issue_weights = [&](const Item& it, int ph, int ring) {
        const char* wsrc = reinterpret_cast<const char*>(P.w + ((size_t)it.a * P.NCB + it.cb) * nph * W_REC);
#pragma unroll
        for (int i = 0; i < W_PW; ++i) {
            const char* base = wsrc + ((size_t)ph * W_REC + (wave + 8 * i) * 64) * 16;
            dma16(base, lane16, w_l + ring * W_REC + (wave + 8 * i) * 64);
        }
    };
    const unsigned lane16h = (lane & 31) * 16;
    auto issue_consts = [&](const Item& it, int par) {
        if (wave == 0 && P.bias) dma16(reinterpret_cast<const char*>(P.bias + it.cb * (MT * 32)), lane16h, ec_l + par * EC_REC);
        if ((wave == 1 || wave == 2) && P.yrec && P.coef)
            dma16(reinterpret_cast<const char*>(P.coef + ((size_t)it.b * 2 + (wave - 1)) * P.Cout + it.cb * (MT * 32)), lane16h,
                  ec_l + par * EC_REC + wave * 64);
    };

    bf16x8 fw[2][MW][2];     // [set][m][hl]   weight tiles of one combo-step
    bf16x8 fx[2][NROW][2];   // [set][n][hl]   input rows of one column shift
    const int wfrag = wm * MW * 64 + lane;
    auto load_fw = [&](int set, int ring, int c) {
        const int bb = c >> 1, v = ((c + 1) >> 1) - bb;           // c: 0 -> (0, 0), 1 -> (0, 1), 2 -> (1, 0), 3 -> (1, 1)
        const u32x4* wst = w_l + ring * W_REC + wfrag;
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fw[set][m][hl] = __builtin_bit_cast(bf16x8, wst[(((hl * 2 + bb) * 2 + v) * MT + m) * 64]);
    };
    auto load_fx = [&](int set, int xfrag, int stage, int u, int s) {
        const u32x4* ist = in_l + stage * IS::PAD + xfrag + u * COLS + s;
#pragma unroll
        for (int n = 0; n < NROW; ++n)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fx[set][n][hl] = __builtin_bit_cast(bf16x8, ist[hl * IS::HALF_PAD + n * COLS]);
    };

    Item cur, nxt;
    int work = next_valid(blockIdx.x, cur);
    if (work >= total) return;
    unsigned ioff[IS::PW];
    make_ioff(cur, ioff);
    issue_input(cur, ioff, 0, 0);
    issue_weights(cur, 0, 0);
    issue_weights(cur, 1, 1);
    issue_consts(cur, 0);
    stagger_start(P, wave);
    int par = 0, r0 = 0;       // constants-buffer parity, ring slot of this item's chunk 0

    while (true) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int xfrag = (kg * ROWS + wr * NROW + cur.a) * COLS + l31;   // halo row of output row n at tap row u: + (n + u)*COLS
        int rs[3];                                                        // ring slot of local phase p: rs[p % 3]
#pragma unroll
        for (int i = 0; i < 3; ++i) rs[i] = (r0 + i) % 3;
        load_fw(0, rs[0], 0);
        load_fx(0, xfrag, 0, 0, 0);
        const int work_n = next_valid(work + gridDim.x, nxt);
        unsigned ioff_n[IS::PW];
        if (work_n < total) make_ioff(nxt, ioff_n);
        const int r0_n = (r0 + nph) % 3;                                  // = (slot of the last chunk + 1) % 3

        f32x16 acc[MW][NROW][2];   // [m][n][bb]
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int n = 0; n < NROW; ++n)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[m][n][bb][q] = 0.0f;

        // one trip = 3 K-steps = 6 phases = 24 combo-steps: the register sets are compile-time (fw set = combo-step parity; fx set =
        // parity of the running shift counter 3*phase + s) and so is the ring index modulo the item's r0.  NK % 3 != 0: the surplus
        // K-steps of the last trip are skipped (wave-uniform branch).
        for (int k3 = 0; k3 < P.NK; k3 += 3) {
#pragma unroll
            for (int t = 0; t < 24; ++t) {
                const int pl_ = t >> 2, c = t & 3;                 // local phase 0..5, combo-step
                const int kk = pl_ >> 1, u = pl_ & 1, s = (c + 1) >> 1, bb = c >> 1;
                const int k = k3 + kk, ph = k * 2 + u;
                const int ws = t & 1, xs = (3 * pl_ + s) & 1;
                if (k < P.NK) {
                    MDT_PIN();
                    if (c < 3) {
                        load_fw(ws ^ 1, rs[pl_ % 3], c + 1);
                        if (c != 1) load_fx(xs ^ 1, xfrag, k & 1, u, s + 1);
                    } else {
                        const int pl1 = (pl_ + 1) % 6, kk1 = pl1 >> 1, u1 = pl1 & 1;
                        const int k1 = (pl_ < 5 ? k3 : k3 + 3) + kk1;
                        if (k1 < P.NK) {
                            load_fw(ws ^ 1, rs[pl1 % 3], 0);
                            load_fx(xs ^ 1, xfrag, k1 & 1, u1, 0);
                        }
                    }
                    MDT_PIN();
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int n = 0; n < NROW; ++n)
#pragma unroll
                            for (int m = 0; m < MW; ++m)
                                acc[m][n][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ws][m][term == 0 ? 1 : 0], fx[xs][n][term == 1 ? 1 : 0],
                                                                                        acc[m][n][bb], 0, 0, 0);
                    MDT_PIN();
                    if (c == 1) {
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                    // staggered DMA issue (see k_conv3x3_rec): waves 0-3 behind the barrier, waves 4-7 one combo-step later
                    if ((c == 1 && wave < 4) || (c == 2 && wave >= 4)) {
                        if (ph + 2 < nph) issue_weights(cur, ph + 2, rs[(pl_ + 2) % 3]);
                        if (u == 0 && k + 1 < P.NK) issue_input(cur, ioff, k + 1, (k + 1) & 1);
                        if (ph + 1 == nph && work_n < total) {
                            // last phase: the two ring slots it does not read and input stage 0 (NK is even: the last K-step sits in
                            // stage 1) take the next item's first operands
                            issue_input(nxt, ioff_n, 0, 0);
                            issue_weights(nxt, 0, r0_n);
                            issue_weights(nxt, 1, (r0_n + 1) % 3);
                            issue_consts(nxt, par ^ 1);
                        }
                    }
                }
            }
        }

        EpiCtx E;
        E.res = P.res; E.y32 = P.y32; E.yrec = P.yrec;
        E.has_bias = P.bias != nullptr; E.has_act = P.yrec != nullptr && P.coef != nullptr;
        E.Cout = P.Cout; E.H = P.H; E.W = P.W; E.b = cur.b; E.kg = kg;
        E.HW = (size_t)P.H * P.W; E.planeO = (size_t)(P.H + 2) * rec_pitch(P.W); E.WpO = rec_pitch(P.W); E.dbg = pdbg(P.dbg);
        const int xi = cur.x0 + l31;
        int ys[NROW];
#pragma unroll
        for (int n = 0; n < NROW; ++n) {
            const int yi = cur.y0 + wr * NROW + n;
            ys[n] = yi < P.Hin ? 2 * yi + cur.a : P.H;      // rows past the input's last row: marked invalid
        }
        if constexpr (ST) {      // unit = (input pixel tile, row parity, row group of the block): conv_stats_finish_launch(units = ptiles * 2 * 4)
            const int ptile = (cur.y0 / TH) * P.PX + cur.x0 / 32;
            E.st = P.gn_part + ((((((size_t)cur.b * P.ptiles + ptile) * 2 + cur.a) * (8 / WM) + wr) * P.NCB + cur.cb) * (MT * 8) + wm * MW * 8) * 2;
        }
        if (!(pdbg(P.dbg) & 1)) {
            epilogue_item<2, NROW, MW, 64, ST>(E, ec_l + par * EC_REC, acc, wm * MW, cur.cb * MT + wm * MW, ys, 2 * xi, xi < P.Win, ResRows<NROW>{});
        }
        if (work_n >= total) break;
        work = work_n;
        cur = nxt;
        par ^= 1;
        r0 = r0_n;
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) ioff[i] = ioff_n[i];
    }
}

