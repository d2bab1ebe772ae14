// wire_io.cpp - the reference's serialization formats over the flat layouts (include/tfhe_hip_io.h).
// Host code only.  Format facts are cited from the reference (src/libtfhe/tfhe_io.cpp,
// src/libtfhe/tfhe_generic_streams.cpp); nothing here is needed on the hot path itself.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "internal.h"
#include "tfhe_hip_io.h"

namespace {

constexpr int kLwe = TFHE_HIP_n, kN = TFHE_HIP_N, kKpl = TFHE_HIP_KPL, kT = TFHE_HIP_KS_T, kBase = TFHE_HIP_KS_BASE;
constexpr int32_t kUidSample = 42, kUidKs = 200, kUidBk = 201;  // tfhe_generic_streams.h:15,29,30
constexpr size_t kKsElems = (size_t) kN * kT * kBase * (kLwe + 1);
constexpr size_t kBkElems = (size_t) kLwe * kKpl * 2 * kN;

int bad(const std::string &msg) { return tfhe_hip_internal_fail(TFHE_HIP_ERR_ARG, msg.c_str()); }

// ---- reading ----
struct Cursor {
    const unsigned char *p;
    size_t len, pos;
    bool eof() const { return pos >= len; }
    // CIstream::getLine (tfhe_generic_streams.cpp:72-79): drops '\r', stops after '\n'
    bool line(std::string &out) {
        out.clear();
        if (eof()) return false;
        while (pos < len) {
            const unsigned char c = p[pos++];
            if (c == '\r') continue;
            if (c == '\n') return true;
            out.push_back((char) c);
        }
        return true;
    }
    bool bytes(void *dst, size_t n) {
        if (len - pos < n) return false;
        memcpy(dst, p + pos, n);
        pos += n;
        return true;
    }
};

// one "-----BEGIN X-----" ... "-----END X-----" section (new_TextModeProperties_fromIstream, :119-154): lines before
// the BEGIN line and lines without ": " are skipped
bool read_section(Cursor &c, std::string &title, std::map<std::string, std::string> &props) {
    std::string ln, end;
    bool started = false;
    props.clear();
    while (c.line(ln)) {
        const size_t n = ln.size();
        if (n >= 16 && ln.compare(0, 11, "-----BEGIN ") == 0 && ln.compare(n - 5, 5, "-----") == 0) {
            title = ln.substr(11, n - 16);
            end = "-----END " + title + "-----";
            started = true;
            continue;
        }
        if (!started) continue;
        if (ln == end) return true;
        const size_t pos = ln.find(": ");
        if (pos == std::string::npos) continue;
        props[ln.substr(0, pos)] = ln.substr(pos + 2);
    }
    return false;
}

int expect_section(Cursor &c, const char *title, std::initializer_list<std::pair<const char *, long>> ints) {
    std::string t;
    std::map<std::string, std::string> props;
    if (!read_section(c, t, props)) return bad(std::string("truncated or missing text section ") + title);
    if (t != title) return bad("expected section " + std::string(title) + ", found " + t);
    for (const auto &kv : ints) {
        auto it = props.find(kv.first);
        if (it == props.end()) return bad(std::string(title) + ": property " + kv.first + " missing");
        char *endp = nullptr;
        const long v = strtol(it->second.c_str(), &endp, 10);
        if (endp == it->second.c_str()) return bad(std::string(title) + ": property " + kv.first + " is not a number");
        if (v != kv.second)
            return bad(std::string(title) + ": " + kv.first + " = " + std::to_string(v) + ", this build supports only " +
                       std::to_string(kv.second) + " (this library is built for the " + std::to_string(TFHE_HIP_LAMBDA) + "-bit parameter set)");
    }
    return TFHE_HIP_OK;
}

int read_params(Cursor &c) {  // read_new_tfheGateBootstrappingParameters, tfhe_io.cpp:1037-1045
    if (int r = expect_section(c, "GATEBOOTSPARAMS", {{"ks_t", kT}, {"ks_basebit", 2}})) return r;
    if (int r = expect_section(c, "LWEPARAMS", {{"n", kLwe}})) return r;
    if (int r = expect_section(c, "TLWEPARAMS", {{"N", kN}, {"k", 1}})) return r;
    return expect_section(c, "TGSWPARAMS", {{"l", TFHE_HIP_L}, {"Bgbit", TFHE_HIP_BGBIT}});
}

int read_blob(Cursor &c, int32_t uid, const char *what, double *variance, int32_t *dst, size_t elems) {
    int32_t got = -1;
    double var = -1;
    if (!c.bytes(&got, 4)) return bad(std::string(what) + ": truncated");
    if (got != uid) return bad(std::string(what) + ": wrong type id " + std::to_string(got));
    if (!c.bytes(&var, 8) || !c.bytes(dst, elems * sizeof(int32_t))) return bad(std::string(what) + ": truncated");
    if (variance) *variance = var;
    return TFHE_HIP_OK;
}

// ---- writing ----
struct Sink {
    unsigned char *p;
    size_t cap, pos;
    bool ok = true;
    void put(const void *src, size_t n) {
        if (!ok || cap - pos < n) {
            ok = false;
            return;
        }
        memcpy(p + pos, src, n);
        pos += n;
    }
    void text(const std::string &s) { put(s.data(), s.size()); }
};
std::string fmt_int(long v) {  // setProperty_int64_t, tfhe_generic_streams.cpp:52-56
    char b[64];
    snprintf(b, sizeof b, "%10" PRId64, (int64_t) v);
    return b;
}
std::string fmt_dbl(double v) {  // setProperty_double, :46-50
    char b[64];
    snprintf(b, sizeof b, "%.8lf", v);
    return b;
}
// print_TextModeProperties_toOStream (:166-173): std::map order = byte-wise lexicographic
void write_section(Sink &s, const char *title, const std::map<std::string, std::string> &props) {
    s.text(std::string("-----BEGIN ") + title + "-----\n");
    for (const auto &kv : props) s.text(kv.first + ": " + kv.second + "\n");
    s.text(std::string("-----END ") + title + "-----\n");
}
#if TFHE_HIP_LAMBDA == 128  // default_128bit_gate_bootstrapping_parameters, tfhe_gate_bootstrapping.cpp:51-75
const double kKsStdev = 1.0 / 32768.0, kBkStdev = 1.0 / 33554432.0, kMaxStdev = 0.012467;
#else                       // default_80bit_gate_bootstrapping_parameters, :23-48
const double kKsStdev = 2.44e-5, kBkStdev = 7.18e-9, kMaxStdev = 0.012467;
#endif
void write_params(Sink &s) {  // write_tfheGateBootstrappingParameters, tfhe_io.cpp:1031-1035
    write_section(s, "GATEBOOTSPARAMS", {{"ks_t", fmt_int(kT)}, {"ks_basebit", fmt_int(2)}});
    write_section(s, "LWEPARAMS", {{"n", fmt_int(kLwe)}, {"alpha_min", fmt_dbl(kKsStdev)}, {"alpha_max", fmt_dbl(kMaxStdev)}});
    write_section(s, "TLWEPARAMS", {{"N", fmt_int(kN)}, {"k", fmt_int(1)}, {"alpha_min", fmt_dbl(kBkStdev)},
                                    {"alpha_max", fmt_dbl(kMaxStdev)}});
    write_section(s, "TGSWPARAMS", {{"l", fmt_int(TFHE_HIP_L)}, {"Bgbit", fmt_int(TFHE_HIP_BGBIT)}});
}

struct File {
    FILE *f = nullptr;
    ~File() {
        if (f) fclose(f);
    }
};

int slurp(const char *path, std::vector<unsigned char> &out) {
    if (!path) return bad("null path");
    File F;
    F.f = fopen(path, "rb");
    if (!F.f) return bad(std::string("cannot open ") + path);
    if (fseek(F.f, 0, SEEK_END) != 0) return bad(std::string("cannot seek ") + path);
    const long sz = ftell(F.f);
    if (sz < 0) return bad(std::string("cannot size ") + path);
    rewind(F.f);
    out.resize((size_t) sz);
    if (sz && fread(out.data(), 1, (size_t) sz, F.f) != (size_t) sz) return bad(std::string("short read on ") + path);
    return TFHE_HIP_OK;
}

}  // namespace

extern "C" {

int tfhe_hip_io_read_params(const void *buf, size_t len, size_t *consumed) {
    if (!buf) return bad("null buffer");
    Cursor c{(const unsigned char *) buf, len, 0};
    if (int r = read_params(c)) return r;
    if (consumed) *consumed = c.pos;
    return TFHE_HIP_OK;
}

int tfhe_hip_io_write_params(void *buf, size_t cap, size_t *written) {
    if (!buf) return bad("null buffer");
    Sink s{(unsigned char *) buf, cap, 0};
    write_params(s);
    if (!s.ok) return bad("output buffer too small");
    if (written) *written = s.pos;
    return TFHE_HIP_OK;
}

size_t tfhe_hip_io_cloud_keyset_bytes(void) { return 1024 + 2 * 12 + (kKsElems + kBkElems) * sizeof(int32_t); }

int tfhe_hip_io_read_cloud_keyset(const void *buf, size_t len, int32_t *bk_coef, int32_t *ks, double *variances,
                                  size_t *consumed) {
    if (!buf || !bk_coef || !ks) return bad("null buffer");
    Cursor c{(const unsigned char *) buf, len, 0};
    if (int r = read_params(c)) return r;
    // read_new_lweBootstrappingKey with both parameter objects given (tfhe_io.cpp:950-970): keyswitch section first
    if (int r = expect_section(c, "LWEKSPARAMS", {{"n", kN}, {"t", kT}, {"basebit", 2}})) return r;
    if (int r = read_blob(c, kUidKs, "keyswitch key", variances ? variances + 0 : nullptr, ks, kKsElems)) return r;
    if (int r = read_blob(c, kUidBk, "bootstrapping key", variances ? variances + 1 : nullptr, bk_coef, kBkElems))
        return r;
    if (consumed) *consumed = c.pos;
    return TFHE_HIP_OK;
}

int tfhe_hip_io_write_cloud_keyset(void *buf, size_t cap, size_t *written, const int32_t *bk_coef, const int32_t *ks,
                                   const double *variances) {
    if (!buf || !bk_coef || !ks) return bad("null buffer");
    Sink s{(unsigned char *) buf, cap, 0};
    write_params(s);  // write_tfheGateBootstrappingCloudKeySet, tfhe_io.cpp:1099-1103
    write_section(s, "LWEKSPARAMS", {{"n", fmt_int(kN)}, {"t", fmt_int(kT)}, {"basebit", fmt_int(2)}});
    const double vks = variances ? variances[0] : 0.0, vbk = variances ? variances[1] : 0.0;
    s.put(&kUidKs, 4);
    s.put(&vks, 8);
    s.put(ks, kKsElems * sizeof(int32_t));
    s.put(&kUidBk, 4);
    s.put(&vbk, 8);
    s.put(bk_coef, kBkElems * sizeof(int32_t));
    if (!s.ok) return bad("output buffer too small");
    if (written) *written = s.pos;
    return TFHE_HIP_OK;
}

int tfhe_hip_io_read_cloud_keyset_file(const char *path, int32_t *bk_coef, int32_t *ks, double *variances) {
    std::vector<unsigned char> data;
    if (int r = slurp(path, data)) return r;
    return tfhe_hip_io_read_cloud_keyset(data.data(), data.size(), bk_coef, ks, variances, nullptr);
}

int tfhe_hip_io_write_cloud_keyset_file(const char *path, const int32_t *bk_coef, const int32_t *ks,
                                        const double *variances) {
    if (!path) return bad("null path");
    std::vector<unsigned char> data(tfhe_hip_io_cloud_keyset_bytes());
    size_t n = 0;
    if (int r = tfhe_hip_io_write_cloud_keyset(data.data(), data.size(), &n, bk_coef, ks, variances)) return r;
    File F;
    F.f = fopen(path, "wb");
    if (!F.f) return bad(std::string("cannot create ") + path);
    if (fwrite(data.data(), 1, n, F.f) != n) return bad(std::string("short write on ") + path);
    return TFHE_HIP_OK;
}

int tfhe_hip_load_cloud_keyset_buffer(tfhe_hip_ctx *ctx, const void *buf, size_t len) {
    if (!ctx) return bad("null context");
    std::vector<int32_t> bk(kBkElems), ks(kKsElems);
    if (int r = tfhe_hip_io_read_cloud_keyset(buf, len, bk.data(), ks.data(), nullptr, nullptr)) return r;
    return tfhe_hip_load_keys_coef(ctx, bk.data(), ks.data());
}

int tfhe_hip_load_cloud_keyset_file(tfhe_hip_ctx *ctx, const char *path) {
    if (!ctx) return bad("null context");
    std::vector<unsigned char> data;
    if (int r = slurp(path, data)) return r;
    return tfhe_hip_load_cloud_keyset_buffer(ctx, data.data(), data.size());
}

int tfhe_hip_io_read_samples(const void *buf, size_t len, int32_t *samples, double *variances, int64_t count,
                             size_t *consumed) {
    if (count < 0 || !buf || (count > 0 && !samples)) return bad("bad buffer or count");
    Cursor c{(const unsigned char *) buf, len, 0};
    for (int64_t i = 0; i < count; i++) {  // read_lweSample, tfhe_io.cpp:90-98
        int32_t uid = -1;
        double var = 0;
        if (!c.bytes(&uid, 4)) return bad("ciphertext " + std::to_string(i) + ": truncated");
        if (uid != kUidSample) return bad("ciphertext " + std::to_string(i) + ": wrong type id " + std::to_string(uid));
        if (!c.bytes(samples + i * (kLwe + 1), (kLwe + 1) * sizeof(int32_t)) || !c.bytes(&var, 8))
            return bad("ciphertext " + std::to_string(i) + ": truncated");
        if (variances) variances[i] = var;
    }
    if (consumed) *consumed = c.pos;
    return TFHE_HIP_OK;
}

int tfhe_hip_io_write_samples(void *buf, size_t cap, size_t *written, const int32_t *samples, const double *variances,
                              int64_t count) {
    if (count < 0 || !buf || (count > 0 && !samples)) return bad("bad buffer or count");
    Sink s{(unsigned char *) buf, cap, 0};
    for (int64_t i = 0; i < count; i++) {  // write_lweSample, tfhe_io.cpp:101-107
        const double var = variances ? variances[i] : 0.0;
        s.put(&kUidSample, 4);
        s.put(samples + i * (kLwe + 1), (kLwe + 1) * sizeof(int32_t));
        s.put(&var, 8);
    }
    if (!s.ok) return bad("output buffer too small");
    if (written) *written = s.pos;
    return TFHE_HIP_OK;
}

}  // extern "C"
