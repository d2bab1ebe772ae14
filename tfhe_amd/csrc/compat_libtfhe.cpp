// Struct-level libtfhe shim (include/tfhe_hip_compat.h): flattens the reference's pointer-rich structs, calls the
// batched C-ABI, scatters results back.  Host code only; no kernels here.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <memory>
#include <mutex>
#include <vector>
#include "../../include/tfhe_hip_compat.h"
#include "internal.h"

namespace {
constexpr int kn = TFHE_HIP_n, kN = TFHE_HIP_N, kKpl = TFHE_HIP_KPL, kT = TFHE_HIP_KS_T, kBase = TFHE_HIP_KS_BASE;

std::mutex g_mu;
tfhe_hip_group *g_group = nullptr;       // the default group: $TFHE_HIP_DEVICES = "0,1,..." (or $TFHE_HIP_DEVICE, default 0)
const void *g_loaded_keyset = nullptr;   // TFheGateBootstrappingCloudKeySet* or TGswSampleFFT* currently uploaded
uint64_t g_loaded_print = 0;             // and the fingerprint of its contents at upload time

[[noreturn]] void die(const char *what) {
    std::fprintf(stderr, "tfhe_hip: %s: %s\n", what, tfhe_hip_last_error());
    std::abort();
}

// Content fingerprint of a key (FNV-1a over a sample of it: the first two spectrum values of each of the n*2l*2
// polynomials of bkFFT and the b of every 8th keyswitch sample).  The shim keeps the uploaded key across calls - the
// reference re-copies the whole bkFFT on every call (lwe-bootstrapping-functions-fft.cpp:112-133) - and must notice a
// different key behind the same pointer (a keyset freed and another allocated at its address, bkFFT rewritten in place).
uint64_t key_fingerprint(const TGswSampleFFT *bk, const LweKeySwitchKey *ks) {
    uint64_t h = 1469598103934665603ULL;
    auto mix = [&](const void *p, size_t nbytes) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < nbytes; i++) {
            h ^= b[i];
            h *= 1099511628211ULL;
        }
    };
    for (int i = 0; i < kn; i++)
        for (int p = 0; p < kKpl; p++)
            for (int c = 0; c < 2; c++) mix(bk[i].all_samples[p].a[c].coefsC, 4 * sizeof(double));
    if (ks)
        for (long r = 0; r < (long) kN * kT * kBase; r += 8) {
            mix(&ks->ks0_raw[r].b, sizeof(int32_t));
            mix(ks->ks0_raw[r].a, 2 * sizeof(int32_t));
        }
    return h;
}

void flatten_bkfft(std::vector<double> &dst, const TGswSampleFFT *bk) {
    dst.resize((size_t) kn * kKpl * 2 * kN);
    for (int i = 0; i < kn; i++)
        for (int p = 0; p < kKpl; p++)
            for (int c = 0; c < 2; c++)
                std::memcpy(dst.data() + (((size_t) i * kKpl + p) * 2 + c) * kN, bk[i].all_samples[p].a[c].coefsC,
                            sizeof(double) * kN);
}

void flatten_keys(std::vector<double> &fbk, std::vector<int32_t> &fks, const TGswSampleFFT *bk, const LweKeySwitchKey *ks) {
    flatten_bkfft(fbk, bk);
    fks.assign((size_t) kN * kT * kBase * (kn + 1), 0);
    if (ks) {
        for (int i = 0; i < kN; i++)
            for (int j = 0; j < kT; j++)
                for (int d = 0; d < kBase; d++) {
                    const LweSample *s = &ks->ks[i][j][d];
                    int32_t *row = fks.data() + (((size_t) i * kT + j) * kBase + d) * (kn + 1);
                    std::memcpy(row, s->a, sizeof(int32_t) * kn);
                    row[kn] = s->b;
                }
    }
}
bool ks_shape_ok(const LweKeySwitchKey *ks) { return !ks || (ks->n == kN && ks->t == kT && ks->base == kBase); }

int load_keys(tfhe_hip_ctx *ctx, const TGswSampleFFT *bk, const LweKeySwitchKey *ks) {
    if (!ks_shape_ok(ks)) return TFHE_HIP_ERR_ARG;
    std::vector<double> fbk;
    std::vector<int32_t> fks;
    flatten_keys(fbk, fks, bk, ks);
    return tfhe_hip_load_keys_fft(ctx, fbk.data(), fks.data());
}

tfhe_hip_group *default_group_locked() {
    if (!g_group) {
        std::vector<int> devs;
        if (const char *list = std::getenv("TFHE_HIP_DEVICES")) {
            for (const char *p = list; *p;) {
                char *end = nullptr;
                const long v = std::strtol(p, &end, 10);
                if (end == p) break;
                devs.push_back((int) v);
                p = *end == ',' ? end + 1 : end;
            }
        }
        if (devs.empty()) {
            const char *d = std::getenv("TFHE_HIP_DEVICE");
            devs.push_back(d ? std::atoi(d) : 0);
        }
        if (tfhe_hip_group_create(&g_group, devs.data(), (int) devs.size()) != 0) die("creating the default device group");
    }
    return g_group;
}

// uploads `bk` (+ `ks`) to every member of the default group unless that very key - same pointer, same contents - is there
void ensure_keys_locked(tfhe_hip_group *grp, const void *owner, const TGswSampleFFT *bk, const LweKeySwitchKey *ks) {
    const uint64_t print = key_fingerprint(bk, ks);
    if (g_loaded_keyset == owner && g_loaded_print == print) return;
    if (!ks_shape_ok(ks)) die("keyswitch key of another parameter set");
    std::vector<double> fbk;
    std::vector<int32_t> fks;
    flatten_keys(fbk, fks, bk, ks);
    if (tfhe_hip_group_load_keys_fft(grp, fbk.data(), fks.data()) != 0) die("uploading the keys");
    g_loaded_keyset = owner;
    g_loaded_print = print;
}

struct SamplesJob {
    int gate;
    LweSample *result;
    const LweSample *a, *b, *c;
    int32_t count;
};
int samples_member(void *user, tfhe_hip_ctx *ctx, int member, int members) {  // member's contiguous shard of the arrays
    const SamplesJob &j = *static_cast<const SamplesJob *>(user);
    const long long base = j.count / members, rem = j.count % members;
    const long long lo = member * base + std::min<long long>(member, rem), n = base + (member < rem ? 1 : 0);
    if (n == 0) return TFHE_HIP_OK;
    return tfhe_hip_gate_batch_samples(ctx, j.gate, j.result + lo, j.a ? j.a + lo : nullptr, j.b ? j.b + lo : nullptr,
                                       j.c ? j.c + lo : nullptr, (int32_t) n);
}

void boots_batch(int gate, LweSample *result, const LweSample *a, const LweSample *b, const LweSample *c, int32_t count,
                 const TFheGateBootstrappingCloudKeySet *bk) {
    std::lock_guard<std::mutex> lock(g_mu);
    tfhe_hip_group *grp = default_group_locked();
    if (gate <= TFHE_HIP_MUX) {
        if (!bk || !bk->bkFFT || !bk->bkFFT->bkFFT || !bk->bkFFT->ks) die("null cloud keyset");
        const LweBootstrappingKeyFFT *k = bk->bkFFT;
        if (k->in_out_params->n != kn || k->accum_params->N != kN || k->accum_params->k != 1 ||
            k->bk_params->l != TFHE_HIP_L || k->bk_params->Bgbit != TFHE_HIP_BGBIT)
            die("cloud keyset of another parameter set");
        ensure_keys_locked(grp, bk, k->bkFFT, k->ks);
    }
    if (tfhe_hip_group_gate_batch_samples(grp, gate, result, a, b, c, count) != 0) die("gate batch");
}
}  // namespace

extern "C" {

int tfhe_hip_load_cloud_keyset(tfhe_hip_ctx *ctx, const TFheGateBootstrappingCloudKeySet *keyset) {
    if (!ctx || !keyset || !keyset->bkFFT || !keyset->bkFFT->bkFFT || !keyset->bkFFT->ks) return TFHE_HIP_ERR_ARG;
    const LweBootstrappingKeyFFT *k = keyset->bkFFT;
    if (k->in_out_params->n != kn || k->accum_params->N != kN || k->accum_params->k != 1 || k->bk_params->l != TFHE_HIP_L ||
        k->bk_params->Bgbit != TFHE_HIP_BGBIT)
        return TFHE_HIP_ERR_ARG;   // only the parameter set this library was built for (include/tfhe_hip.h)
    return load_keys(ctx, k->bkFFT, k->ks);
}

// The pointer-rich LweSample arrays (each sample owns a separately allocated a[n], lwesamples.cpp:5-10) are packed
// into flat buffers chunk by chunk through the hooks of the host-buffer pipeline (internal.h): the calling thread packs
// chunk k+1 and scatters the results of chunk k-1 while the GPU works on chunk k, so that the packing (165 MB of small
// copies per 16,384 NAND gates) hides behind the kernels.
int tfhe_hip_gate_batch_samples(tfhe_hip_ctx *ctx, int gate, LweSample *result, const LweSample *a,
                                const LweSample *b, const LweSample *c, int32_t count) {
    if (!ctx || count < 0 || (count > 0 && !result)) return TFHE_HIP_ERR_ARG;
    if (count == 0) return TFHE_HIP_OK;
    // flat staging buffers, kept per thread and only ever grown: a fresh 165 MB per call would cost more in page
    // faults and zero-filling than the packing itself
    struct Staging {
        std::unique_ptr<int32_t[]> p;
        size_t cap = 0;
        int32_t *get(size_t elems) {
            if (elems > cap) {
                p.reset(new int32_t[elems]);
                cap = elems;
            }
            return p.get();
        }
    };
    thread_local Staging st_a, st_b, st_c, st_o;
    struct Job {
        LweSample *result;
        const LweSample *a, *b, *c;
        int32_t *fa, *fb, *fc, *fo;
    };
    const size_t elems = (size_t) count * (kn + 1);
    Job job{result, a, b, c, a ? st_a.get(elems) : nullptr, b ? st_b.get(elems) : nullptr, c ? st_c.get(elems) : nullptr,
            st_o.get(elems)};
    static const auto fill = [](void *user, long long off, long long n) {
        Job &j = *static_cast<Job *>(user);
        auto pr = [&](int32_t *dst, const LweSample *s) {
            for (long long i = off; i < off + n; i++) {
                std::memcpy(dst + (size_t) i * (kn + 1), s[i].a, sizeof(int32_t) * kn);
                dst[(size_t) i * (kn + 1) + kn] = s[i].b;
            }
        };
        if (j.a) pr(j.fa, j.a);
        if (j.b) pr(j.fb, j.b);
        if (j.c) pr(j.fc, j.c);
    };
    static const auto drain = [](void *user, long long off, long long n) {
        Job &j = *static_cast<Job *>(user);
        for (long long i = off; i < off + n; i++) {
            LweSample &r = j.result[i];
            std::memcpy(r.a, j.fo + (size_t) i * (kn + 1), sizeof(int32_t) * kn);
            r.b = j.fo[(size_t) i * (kn + 1) + kn];
            r.current_variance = 0.;   // bookkeeping only in the reference (SURVEY 8(a) note 2)
        }
    };
    return tfhe_hip_internal_gate_batch(ctx, gate, job.fo, job.fa, job.fb, job.fc, count, +fill, +drain, &job);
}

#define BINARY(NAME, ID)                                                                                      \
    void NAME(LweSample *r, const LweSample *ca, const LweSample *cb, int32_t count,                          \
              const TFheGateBootstrappingCloudKeySet *bk) {                                                   \
        boots_batch(ID, r, ca, cb, nullptr, count, bk);                                                       \
    }
BINARY(bootsNAND_batch, TFHE_HIP_NAND)
BINARY(bootsOR_batch, TFHE_HIP_OR)
BINARY(bootsAND_batch, TFHE_HIP_AND)
BINARY(bootsXOR_batch, TFHE_HIP_XOR)
BINARY(bootsXNOR_batch, TFHE_HIP_XNOR)
BINARY(bootsNOR_batch, TFHE_HIP_NOR)
BINARY(bootsANDNY_batch, TFHE_HIP_ANDNY)
BINARY(bootsANDYN_batch, TFHE_HIP_ANDYN)
BINARY(bootsORNY_batch, TFHE_HIP_ORNY)
BINARY(bootsORYN_batch, TFHE_HIP_ORYN)
#undef BINARY

void bootsMUX_batch(LweSample *r, const LweSample *a, const LweSample *b, const LweSample *c, int32_t count,
                    const TFheGateBootstrappingCloudKeySet *bk) {
    boots_batch(TFHE_HIP_MUX, r, a, b, c, count, bk);
}
void bootsNOT_batch(LweSample *r, const LweSample *ca, int32_t count, const TFheGateBootstrappingCloudKeySet *bk) {
    boots_batch(TFHE_HIP_NOT, r, ca, nullptr, nullptr, count, bk);
}
void bootsCOPY_batch(LweSample *r, const LweSample *ca, int32_t count, const TFheGateBootstrappingCloudKeySet *bk) {
    boots_batch(TFHE_HIP_COPY, r, ca, nullptr, nullptr, count, bk);
}
void bootsCONSTANT_batch(LweSample *r, int32_t value, int32_t count, const TFheGateBootstrappingCloudKeySet *bk) {
    boots_batch(value ? TFHE_HIP_CONST1 : TFHE_HIP_CONST0, r, nullptr, nullptr, nullptr, count, bk);
}

void tfhe_hip_blindRotate_FFT(TLweSample *accum, const TGswSampleFFT *bk, const int32_t *bara, const int32_t n,
                              const TGswParams *bk_params) {
    (void) bk_params;
    std::lock_guard<std::mutex> lock(g_mu);
    tfhe_hip_group *grp = default_group_locked();
    if (n != kn || !accum || !bk || !bara) {
        std::fprintf(stderr, "tfhe_hip_blindRotate_FFT: bad arguments\n");
        std::abort();
    }
    ensure_keys_locked(grp, bk, bk, nullptr);
    std::vector<int32_t> acc(2 * kN);
    for (int c = 0; c < 2; c++) std::memcpy(acc.data() + c * kN, accum->a[c].coefsT, sizeof(int32_t) * kN);
    if (tfhe_hip_blind_rotate_batch(tfhe_hip_group_ctx(grp, 0), acc.data(), bara, 1) != 0) die("blind rotation");
    for (int c = 0; c < 2; c++) std::memcpy(accum->a[c].coefsT, acc.data() + c * kN, sizeof(int32_t) * kN);
}

// tfhe_bootstrap_FFT / tfhe_bootstrap_woKS_FFT (lwe-bootstrapping-functions-fft.cpp:266, :222) on the reference's structs:
// ONE sample on the first GPU of the default group; with_ks == 0 leaves the LWE_N sample (result->a has N entries)
void tfhe_hip_bootstrap_FFT(LweSample *result, const LweBootstrappingKeyFFT *bk, int32_t mu, const LweSample *x, int with_ks) {
    std::lock_guard<std::mutex> lock(g_mu);
    tfhe_hip_group *grp = default_group_locked();
    if (!result || !x || !bk || !bk->bkFFT || !bk->ks) die("tfhe_hip_bootstrap_FFT: null argument");
    if (bk->in_out_params->n != kn || bk->accum_params->N != kN || bk->accum_params->k != 1 ||
        bk->bk_params->l != TFHE_HIP_L || bk->bk_params->Bgbit != TFHE_HIP_BGBIT)
        die("bootstrapping key of another parameter set");
    ensure_keys_locked(grp, bk, bk->bkFFT, bk->ks);
    std::vector<int32_t> in(kn + 1), out(kN + 1);
    std::memcpy(in.data(), x->a, sizeof(int32_t) * kn);
    in[kn] = x->b;
    if (tfhe_hip_bootstrap_batch(tfhe_hip_group_ctx(grp, 0), out.data(), in.data(), mu, 1, with_ks) != 0) die("bootstrap");
    const int dim = with_ks ? kn : kN;
    std::memcpy(result->a, out.data(), sizeof(int32_t) * dim);
    result->b = out[dim];
    result->current_variance = 0.;
}

int tfhe_hip_group_gate_batch_samples(tfhe_hip_group *group, int gate, LweSample *result, const LweSample *a,
                                      const LweSample *b, const LweSample *c, int32_t count) {
    if (!group || count < 0 || (count > 0 && !result)) return TFHE_HIP_ERR_ARG;
    if (count == 0) return TFHE_HIP_OK;
    SamplesJob job{gate, result, a, b, c, count};
    return tfhe_hip_internal_group_run(group, samples_member, &job);
}
int tfhe_hip_group_load_cloud_keyset(tfhe_hip_group *group, const TFheGateBootstrappingCloudKeySet *keyset) {
    if (!group || !keyset || !keyset->bkFFT || !keyset->bkFFT->bkFFT || !keyset->bkFFT->ks) return TFHE_HIP_ERR_ARG;
    const LweBootstrappingKeyFFT *k = keyset->bkFFT;
    if (k->in_out_params->n != kn || k->accum_params->N != kN || k->accum_params->k != 1 || k->bk_params->l != TFHE_HIP_L ||
        k->bk_params->Bgbit != TFHE_HIP_BGBIT || !ks_shape_ok(k->ks))
        return TFHE_HIP_ERR_ARG;
    std::vector<double> fbk;
    std::vector<int32_t> fks;
    flatten_keys(fbk, fks, k->bkFFT, k->ks);
    return tfhe_hip_group_load_keys_fft(group, fbk.data(), fks.data());
}

void tfhe_hip_default_context_reset(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_group) tfhe_hip_group_destroy(g_group);
    g_group = nullptr;
    g_loaded_keyset = nullptr;
    g_loaded_print = 0;
}

}  // extern "C"
