// Body of k_conv3x3_rec / k_conv3x3_rec_st (csrc/vae_conv_rec.hip includes this file twice).  The SAME text compiled under two kernel names:
//   MDT_REC_KERNEL = k_conv3x3_rec,    MDT_REC_ST = 0 : the shipping kernel -- its code is what it was before the statistics variant existed
//                                                      (a shared __device__ body template changed hipcc's code for it: 8642 -> 8491 instructions)
//   MDT_REC_KERNEL = k_conv3x3_rec_st, MDT_REC_ST = 1 : + the GroupNorm statistics of the output (epilogue_item<.., ST = true>): slow mode's pooled sites
template <int MW, int WM, int NROW>
__global__ __launch_bounds__(512, 2) void MDT_REC_KERNEL(const ConvRParams P) {
    constexpr bool ST = MDT_REC_ST != 0;
    constexpr int WR = 8 / WM, TH = WR * NROW, MT = MW * WM, HN = NROW / 2;
    constexpr int ROWS = TH + 2, COLS = 34;
    using IS = InStage<ROWS>;
    constexpr int W_REC = 2 * 3 * MT * 64;              // [hl][dx][mt][lane]
    constexpr int W_DMA = W_REC / 64;
    constexpr int W_PW = (W_DMA + 7) / 8;
    __shared__ u32x4 smem[2 * IS::PAD + 3 * W_REC + 2 * EC_REC];
    u32x4* const in_l = smem;
    u32x4* const w_l = smem + 2 * IS::PAD;
    u32x4* const ec_l = smem + 2 * IS::PAD + 3 * W_REC;     // per-channel epilogue constants, two buffers (item parity)

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wr = wave / WM;
    const int Hp = P.H + 2, Wp = rec_pitch(P.W), Pn = P.Cin >> 3;
    const size_t plane = (size_t)Hp * Wp;

    // work -> (sample, pixel tile, cout block).  Workgroups go to XCDs round-robin (id % 8) and grid % 8 == 0 whenever a block
    // sees more than one item, so `work % 8` is this block's XCD for all its items: all cout blocks of a pixel tile stay on one L2.
    const int per_img = ((P.ptiles + 7) / 8) * 8 * P.NCB, total = per_img * P.B;
    auto decode = [&](int work, WorkItem& it) -> bool {
        it.b = work / per_img;
        const int r = work - it.b * per_img, xcd = r & 7, slot = r >> 3;
        const int ptile = (slot / P.NCB) * 8 + xcd;
        it.cb = slot % P.NCB;
        const int py = ptile / P.PX, px = ptile - py * P.PX;
        it.y0 = py * TH;
        it.x0 = px * 32;
        return ptile < P.ptiles;
    };
    auto next_valid = [&](int work, WorkItem& it) -> int {   // first item >= work (stride grid) that is a real tile, or >= total
        while (work < total && !decode(work, it)) work += gridDim.x;
        return work;
    };

    // input DMA map: wave-instruction di = wave + 8 i covers LDS records [64 di, 64 di + 64) of a stage; hl = di / HALF_DMA
    auto make_ioff = [&](const WorkItem& it, unsigned (&ioff)[IS::PW]) {   // byte offsets inside the (K-step, hl) pair of planes
        int ln = lane;
        asm volatile("" : "+v"(ln));      // the (g, r, c) of a piece are re-derived per item (~10 VALU each): kept across the persistent loop they are
                                          // 15 registers that the epilogue's peak (accumulators + two residual buffers) pushes into scratch
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) {
            const int di = wave + 8 * i;
            int s = (di % IS::HALF_DMA) * 64 + ln;
            if (s >= IS::HALF) s = IS::HALF - 1;            // pad lanes shadow the last record (they land in the pad area)
            const int g = s / (ROWS * COLS), p = s - g * (ROWS * COLS);
            const int r = p / COLS, c = p - r * COLS;
            int pr = it.y0 + r, pc = it.x0 + c;             // padded coordinates (image row y0 + r - 1, image column x0 + c - 1)
            pr = pr < Hp ? pr : Hp - 1;                     // ragged block edge: clamp onto the zero border
            pc = (pc < P.W + 1 ? pc : P.W + 1) + REC_COL0;  // (column of the record image: the left border sits at REC_COL0)
            ioff[i] = (unsigned)(((size_t)g * plane + (size_t)pr * Wp + pc) * 16);
        }
    };
    auto issue_input = [&](const WorkItem& it, const unsigned (&ioff)[IS::PW], int k, int stage) {
        const char* xb = reinterpret_cast<const char*>(P.x + (size_t)it.b * 2 * Pn * plane);
#pragma unroll
        for (int i = 0; i < IS::PW; ++i) {
            const int di = wave + 8 * i;
            if (di < IS::DMA) {
                const char* base = xb + ((size_t)(di / IS::HALF_DMA) * Pn + 2 * (size_t)k) * plane * 16;   // wave-uniform
                dma16(base, ioff[i], in_l + stage * IS::PAD + di * 64);
            }
        }
    };
    const unsigned lane16 = lane * 16;
    auto issue_weights = [&](const WorkItem& it, int ph, int ring) {
        const char* wsrc = reinterpret_cast<const char*>(P.w + (size_t)it.cb * P.NK * 3 * W_REC);
#pragma unroll
        for (int i = 0; i < W_PW; ++i)
            if (wave + 8 * i < W_DMA) {
                const char* base = wsrc + ((size_t)ph * W_REC + (wave + 8 * i) * 64) * 16;
                dma16(base, lane16, w_l + ring * W_REC + (wave + 8 * i) * 64);
            }
    };

    // epilogue constants of an item's BM couts: waves 0 / 1 / 2 fetch bias / a / s (512 B each; the upper lanes repeat the
    // lower ones into the pad half of the 1 KB slot)
    const unsigned lane16h = (lane % (MT * 8)) * 16;      // MT * 32 floats = MT * 8 lanes x 16 B; the other lanes repeat them
    auto issue_consts = [&](const WorkItem& it, int par) {
        if (wave == 0 && P.bias) dma16(reinterpret_cast<const char*>(P.bias + it.cb * (MT * 32)), lane16h, ec_l + par * EC_REC);
        if ((wave == 1 || wave == 2) && P.yrec && P.coef)
            dma16(reinterpret_cast<const char*>(P.coef + ((size_t)it.b * 2 + (wave - 1)) * P.Cout + it.cb * (MT * 32)), lane16h,
                  ec_l + par * EC_REC + wave * 64);
    };

    bf16x8 fw[2][MW][2];   // [set][m][hl]
    bf16x8 fx[2][HN][2];   // [set][row of the half-step][hl]
    const int wfrag = wm * MW * 64 + lane;                       // + ((hl*3 + dx)*MT + m)*64
    const int xfrag = (kg * ROWS + wr * NROW) * COLS + l31;      // + hl*HALF_PAD + (n + dy)*COLS + dx
    auto load_fw = [&](int set, int ring, int dx) {
        const u32x4* wst = w_l + ring * W_REC + wfrag;
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fw[set][m][hl] = __builtin_bit_cast(bf16x8, wst[((hl * 3 + dx) * MT + m) * 64]);
    };
    auto load_fx = [&](int set, int stage, int dy, int dx, int h) {
        const u32x4* ist = in_l + stage * IS::PAD + xfrag + (dy + h * HN) * COLS + dx;
#pragma unroll
        for (int n = 0; n < HN; ++n)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) fx[set][n][hl] = __builtin_bit_cast(bf16x8, ist[hl * IS::HALF_PAD + n * COLS]);
    };

    WorkItem cur, nxt;
    int work = next_valid(blockIdx.x, cur);
    if (work >= total) return;
    unsigned ioff[IS::PW];
    make_ioff(cur, ioff);
    issue_input(cur, ioff, 0, 0);
    issue_weights(cur, 0, 0);
    issue_weights(cur, 1, 1);
    issue_consts(cur, 0);
    stagger_start(P, wave);
    const int nph = P.NK * 3;
    int par = 0;

    // probing (MDTILE_REC_DBG bit 3 + MDTILE_REC_STAMPS=<device address>): block 0 records s_memtime per wave and item at
    //   0 item start (behind the barrier) | 1 K loop done | 2 epilogue code done (stores issued) | 4 vmcnt(0) + barrier of the next item passed
    unsigned long long* const stamps = (pdbg(P.dbg) & 8) && P.census && blockIdx.x == 0 ? reinterpret_cast<unsigned long long*>(P.census) : nullptr;
    int item_no = 0;
    auto stamp = [&](int k) {
        if (stamps && lane == 0 && item_no < 64) stamps[(item_no * 8 + wave) * 8 + k] = __builtin_readcyclecounter();
    };
    // a conv2's residual arrives in the accumulators (conv_rec_common.h: ResRows): the first item's rows are requested here, every later
    // item's by the epilogue of the item before it
    f32x16 acc[MW][NROW][1];
    const bool res_in_acc = P.res != nullptr && !(pdbg(P.dbg) & 1);
    auto res_rows = [&](const WorkItem& it, bool on) {
        ResRows<NROW> R;
        R.on = on; R.b = it.b; R.mt_global0 = it.cb * MT + wm * MW;
#pragma unroll
        for (int n = 0; n < NROW; ++n) R.ys[n] = it.y0 + wr * NROW + n;
        int le = lane;
        asm volatile("" : "+v"(le));
        R.x = it.x0 + (le & 31);
        R.x_ok = R.x < P.W;
        return R;
    };
    if (res_in_acc) {
        const ResRows<NROW> R0 = res_rows(cur, true);
#pragma unroll
        for (int m = 0; m < MW; ++m) residual_into_acc<NROW, MW, NROW>(P.res, P.Cout, (size_t)P.H * P.W, P.H, P.W, kg, R0, m, 0, acc);
    }
    while (true) {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces of the item's first operands have landed (and the residual rows)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (item_no > 0) { --item_no; stamp(4); ++item_no; }
        stamp(0);
        load_fw(0, 0, 0);
        load_fx(0, 0, 0, 0, 0);
        const int work_n = next_valid(work + gridDim.x, nxt);
        unsigned ioff_n[IS::PW];
        if (work_n < total) make_ioff(nxt, ioff_n);      // (outside the unrolled K loop: keeps its body under the unroll budget)

        if (!res_in_acc) {
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int n = 0; n < NROW; ++n)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[m][n][0][q] = 0.0f;
        }

        // one trip = 2 K-steps = 6 phases = 18 steps = 36 half-steps: ring slot (= dy), input stage (= kk) and both register-set
        // parities are compile-time constants inside the unrolled body
        for (int k2 = 0; k2 < P.NK; k2 += 2) {
#pragma unroll
            for (int t = 0; t < 36; ++t) {
                const int kk = t / 18, dy = (t / 6) % 3, dx = (t / 2) % 3, h = t & 1;
                const int k = k2 + kk, ph = k * 3 + dy;
                const int xs = t & 1, ws = (t >> 1) & 1;
                // ---- the NEXT half-step's fragments go out first
                MDT_PIN();
                if (h == 0) {
                    load_fx(xs ^ 1, kk, dy, dx, 1);
                } else if (t < 35) {
                    const int t1 = t + 1, kk1 = t1 / 18, dy1 = (t1 / 6) % 3, dx1 = (t1 / 2) % 3;
                    load_fw(ws ^ 1, dy1, dx1);
                    load_fx(xs ^ 1, kk1, dy1, dx1, 0);
                } else if (k2 + 2 < P.NK) {
                    load_fw(ws ^ 1, 0, 0);
                    load_fx(xs ^ 1, 0, 0, 0, 0);
                }
                MDT_PIN();
                // ---- this half-step's MFMAs: term-major over its accumulators (a dependent MFMA is MW*HN issues away)
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int n = 0; n < HN; ++n)
#pragma unroll
                        for (int m = 0; m < MW; ++m)
                            acc[m][h * HN + n][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ws][m][term == 0 ? 1 : 0], fx[xs][n][term == 1 ? 1 : 0],
                                                                                            acc[m][h * HN + n][0], 0, 0, 0);   // w_lo x_hi, w_hi x_lo, w_hi x_hi
                MDT_PIN();
                if (dx == 0 && h == 1) {
                    // Pieces this wave has in flight, oldest first: weight chunk ph+1 (issued behind the previous barrier) and, at
                    // dy = 1, the input tile of K-step k+1 issued right after it.  The weights are read from the end of this phase
                    // on, the input tile only from the end of the dy = 2 phase: at dy = 1 the IS::PW youngest pieces (every wave
                    // issues exactly that many, IS::DMA % 8 == 0) may stay in flight -- a whole extra phase for their HBM round trip.
                    static_assert(IS::DMA % 8 == 0 && IS::PW == 5, "the counted wait below assumes 5 input pieces per wave");
                    if (dy == 1 && k + 1 < P.NK) __builtin_amdgcn_s_waitcnt(0x0F75);   // vmcnt(5)
                    else __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0)
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                // DMA issue of this phase (~10 scalar / VMEM instructions per piece), staggered between the two waves that share a
                // SIMD (w and w + 4): waves 0-3 right behind the barrier, waves 4-7 one half-step later -- while one of the pair
                // issues its pieces the other one keeps the matrix pipe fed.
                // Ring slot of chunk ph+2 = the one chunk ph-1 held: every wave finished reading it before the barrier.
                if ((dx == 0 && h == 1 && wave < 4) || (dx == 1 && h == 0 && wave >= 4)) {
                    if (ph + 2 < nph) issue_weights(cur, ph + 2, (dy + 2) % 3);
                    if (dy == 0 && k + 1 < P.NK) issue_input(cur, ioff, k + 1, (kk + 1) & 1);
                    if (kk == 1 && dy == 2 && k + 1 == P.NK && work_n < total) {
                        // last phase of the item: ring slots 0 / 1 and input stage 0 are out of use (NK is even) -> the next
                        // item's first operands go there now and land under the remaining MFMAs and the epilogue
                        issue_input(nxt, ioff_n, 0, 0);
                        issue_weights(nxt, 0, 0);
                        issue_weights(nxt, 1, 1);
                        issue_consts(nxt, par ^ 1);
                    }
                }
            }
        }

        stamp(1);
        EpiCtx E;
        E.res = P.res; E.y32 = P.y32; E.yrec = P.yrec;
        E.has_bias = P.bias != nullptr; E.has_act = P.yrec != nullptr && P.coef != nullptr;
        E.Cout = P.Cout; E.H = P.H; E.W = P.W; E.b = cur.b; E.kg = kg;
        E.HW = (size_t)P.H * P.W; E.planeO = plane; E.WpO = Wp;
        E.dbg = pdbg(P.dbg);
        int le = lane;
        asm volatile("" : "+v"(le));      // (re-derived: a separate l31 kept alive through the epilogue goes to scratch)
        const int x = cur.x0 + (le & 31);
        int ys[NROW];
#pragma unroll
        for (int n = 0; n < NROW; ++n) ys[n] = cur.y0 + wr * NROW + n;
        if constexpr (ST) {      // this wave's slots of the item: unit = (pixel tile, row group of the block), conv_stats_finish_launch(units = ptiles * WR)
            const int ptile = (cur.y0 / TH) * P.PX + cur.x0 / 32;
            E.st = P.gn_part + (((((size_t)cur.b * P.ptiles + ptile) * WR + wr) * P.NCB + cur.cb) * (MT * 8) + wm * MW * 8) * 2;
        }
        if (!(pdbg(P.dbg) & 1)) {
            epilogue_item<1, NROW, MW, 64, ST>(E, ec_l + par * EC_REC, acc, wm * MW, cur.cb * MT + wm * MW, ys, x, x < P.W, res_rows(nxt, work_n < total));
        }
        stamp(2);
        ++item_no;
        if (work_n >= total) break;
        work = work_n;
        cur = nxt;
        par ^= 1;
        for (int i = 0; i < IS::PW; ++i) ioff[i] = ioff_n[i];
    }
}

