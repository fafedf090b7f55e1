// Times the one-launch decoder layers of a cached decode step alone (cfg2 shapes: B rows, E=512, L layers) and prints
// the stage timestamps of one team.  Build (tools/exp/build_probes.sh):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCAPDEC_FUSED_TRACE tools/exp/fused_step_probe.hip -o tools/exp/bin/fused_step_probe
// Run: fused_step_probe [B] [L] [Tk] [reps] [pad positions per fourth team]
#include "../../image-captioning_amd/csrc/decode_fused.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

static float* dev_random(size_t n, float scale, std::mt19937& g) {
    std::vector<float> h(n);
    std::uniform_real_distribution<float> d(-scale, scale);
    for (auto& v : h) v = d(g);
    float* p = nullptr;
    if (hipMalloc(&p, n * sizeof(float)) != hipSuccess) return nullptr;
    hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    return p;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1024;
    const int L = argc > 2 ? atoi(argv[2]) : 2;
    const int Tk = argc > 3 ? atoi(argv[3]) : 10;
    const int reps = argc > 4 ? atoi(argv[4]) : 50;
    const int E = 512, max_len = 20;
    const int npad = argc > 5 ? atoi(argv[5]) : 0;      // <PAD> positions (side rows) in every fourth team
    std::mt19937 g(1);
    capdec_fused_args a{};
    const float ws = 0.044f;
    for (int l = 0; l < L; ++l) {
        capdec_fused_layer& f = a.ly[l];
        const float** mats[6] = {&f.wq, &f.wk, &f.wv, &f.wo, &f.w1, &f.w2};
        for (auto m : mats) *m = dev_random((size_t)E * E, ws, g);
        const float** vecs[12] = {&f.bq, &f.bk, &f.bv, &f.bo, &f.b1, &f.b2, &f.n1g, &f.n1b, &f.n2g, &f.n2b, &f.n3g, &f.n3b};
        for (auto v : vecs) *v = dev_random(E, 0.5f, g);
        f.cc = dev_random((size_t)B * E, 1.f, g);
        f.kc = dev_random((size_t)B * max_len * E, 1.f, g);
        f.vc = dev_random((size_t)B * max_len * E, 1.f, g);
    }
    const int RW = CAPDEC_FUSED_ROWS;
    const int nteams = (B + RW - 1) / RW;
    a.L = L; a.B = B; a.max_len = max_len; a.Tk = Tk;
    a.teams = nteams;
    a.cache_bytes = (int)((size_t)B * max_len * E * 4);
    a.cc_div = 1;
    a.x = dev_random((size_t)B * E, 1.f, g);
    const size_t TE = (size_t)nteams * RW * E, SE = (size_t)nteams * 16 * E;
    a.q = dev_random(TE, 1.f, g); a.f_ctx = dev_random(TE, 1.f, g); a.f_y = dev_random(TE, 1.f, g);
    a.f_h = dev_random(TE, 1.f, g); a.f_z = dev_random(TE, 1.f, g);
    a.s_ctx = dev_random(SE, 1.f, g); a.s_y = dev_random(SE, 1.f, g); a.s_h = dev_random(SE, 1.f, g); a.s_z = dev_random(SE, 1.f, g);
    a.emb = dev_random((size_t)16 * E, 1.f, g); a.pe = dev_random((size_t)max_len * E, 1.f, g);
    std::vector<int32_t> tok((size_t)B * (max_len + 1), 5), nrows(1, B), done(B, 0), plist((size_t)nteams * 16, -1);
    for (int t = 0; t < nteams; t += 4) {
        for (int k = 0; k < npad && k < 16; ++k) {
            const int b = t * RW + (k % RW), pos = k % (Tk > 1 ? Tk - 1 : 1);
            plist[t * 16 + k] = b * max_len + pos;
            tok[(size_t)b * (max_len + 1) + pos] = 0;
        }
    }
    int32_t *dtok, *dn, *ddone, *dpl;
    CK(hipMalloc(&dtok, tok.size() * 4)); CK(hipMemcpy(dtok, tok.data(), tok.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dn, 4)); CK(hipMemcpy(dn, nrows.data(), 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ddone, B * 4)); CK(hipMemcpy(ddone, done.data(), B * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dpl, plist.size() * 4)); CK(hipMemcpy(dpl, plist.data(), plist.size() * 4, hipMemcpyHostToDevice));
    a.tok = dtok; a.ldtok = max_len + 1; a.nrows = dn; a.done = ddone; a.plist = dpl;
    const size_t ctr_words = (size_t)nteams * 32 + 32;
    unsigned* ctr;
    CK(hipMalloc(&ctr, ctr_words * 4));
    a.ctr = ctr; a.err = ctr + (size_t)nteams * 32;
    unsigned long long* trace;
    CK(hipMalloc(&trace, 8 * 64 * 8));
    CK(hipMemset(trace, 0, 8 * 64 * 8));
    a.trace = trace; a.trace_team = (nteams / 2) / 4 * 4;
    const unsigned grid = (unsigned)((nteams + 7) / 8) * 64u;
    unsigned* hwid;
    CK(hipMalloc(&hwid, grid * 4));
    CK(hipMemset(hwid, 0xff, grid * 4));
    a.hwid = hwid;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float total = 0.f, best = 1e9f;
    for (int it = 0; it < reps + 5; ++it) {
        CK(hipMemsetAsync(ctr, 0, ctr_words * 4, s));
        CK(hipEventRecord(e0, s));
        if (capdec_launch_fused_layers(a, E, s) != 0) { fprintf(stderr, "launch failed\n"); return 1; }
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 5) { total += ms; best = ms < best ? ms : best; }
    }
    unsigned err = 0;
    CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
    printf("B=%d L=%d Tk=%d side rows %d: %.1f us per launch (best %.1f), err=%u\n", B, L, Tk, npad, total / reps * 1e3f, best * 1e3f, err);
    std::vector<unsigned long long> t(8 * 64);
    CK(hipMemcpy(t.data(), trace, t.size() * 8, hipMemcpyDeviceToHost));
    int n = 0;
    while (n < 64 && t[n]) ++n;
    if (getenv("FUSED_PROBE_HWID")) {
        // which workgroups share a CU: (xcc, se, cu) of every workgroup
        std::vector<unsigned> hw(grid);
        CK(hipMemcpy(hw.data(), hwid, grid * 4, hipMemcpyDeviceToHost));
        for (unsigned i = 0; i < grid; ++i) {
            if (hw[i] == 0xffffffffu) continue;
            const unsigned xg = i & 7, gslot = i >> 3;
            printf("wg %4u team %3u member %u: xcc %u se %u sh %u cu %2u\n", i, (gslot >> 3) * 8 + xg, gslot & 7, hw[i] >> 16,
                   (hw[i] >> 13) & 7, (hw[i] >> 12) & 1, (hw[i] >> 8) & 15);
        }
    }
    printf("stage marks of team %d (us since the member's first mark; 8 members):\n", a.trace_team);
    for (int k = 0; k < n; ++k) {
        printf("%2d:", k);
        for (int m = 0; m < 8; ++m) printf(" %7.2f", (double)(t[m * 64 + k] - t[m * 64]) * 0.01);
        printf("   d=%6.2f\n", k ? (double)(t[k] - t[k - 1]) * 0.01 : 0.0);
    }
    return 0;
}
