// host-only sanitizer run of tfhe_amd/csrc/wire_io.cpp: round trips, every truncation length of the text part,
// random corruption of the first kilobyte
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "tfhe_hip_io.h"
int tfhe_hip_internal_fail(int code, const char *) { return code; }
extern "C" int tfhe_hip_load_keys_coef(tfhe_hip_ctx *, const int32_t *, const int32_t *) { return 0; }
int main() {
    const size_t nbk = (size_t) TFHE_HIP_n * TFHE_HIP_KPL * 2 * TFHE_HIP_N, nks = (size_t) TFHE_HIP_N * 8 * 4 * (TFHE_HIP_n + 1);
    std::vector<int32_t> bk(nbk), ks(nks), bk2(nbk), ks2(nks);
    std::mt19937 rng(1);
    for (auto &v : bk) v = (int32_t) rng();
    for (auto &v : ks) v = (int32_t) rng();
    std::vector<unsigned char> img(tfhe_hip_io_cloud_keyset_bytes());
    size_t n = 0, used = 0;
    double var[2] = {1e-9, 2e-9}, var2[2];
    if (tfhe_hip_io_write_cloud_keyset(img.data(), img.size(), &n, bk.data(), ks.data(), var)) return 1;
    if (tfhe_hip_io_read_cloud_keyset(img.data(), n, bk2.data(), ks2.data(), var2, &used) || used != n) return 2;
    if (bk != bk2 || ks != ks2) return 3;
    if (tfhe_hip_io_write_cloud_keyset(img.data(), n - 1, &used, bk.data(), ks.data(), var) == 0) return 4;  // too small
    int rejected = 0;
    for (size_t len = 0; len < 1200; len++) rejected += tfhe_hip_io_read_cloud_keyset(img.data(), len, bk2.data(), ks2.data(), var2, &used) != 0;
    for (size_t len : {n - 1, n - 4, n / 2}) rejected += tfhe_hip_io_read_cloud_keyset(img.data(), len, bk2.data(), ks2.data(), var2, &used) != 0;
    int weird = 0;
    for (int t = 0; t < 3000; t++) {
        std::vector<unsigned char> c(img.begin(), img.begin() + 2048);
        for (int k = 0; k < 1 + t % 4; k++) c[rng() % 700] = (unsigned char) rng();
        size_t u = 0;
        weird += tfhe_hip_io_read_params(c.data(), c.size(), &u) == 0;
    }
    std::vector<int32_t> s(5 * (TFHE_HIP_n + 1)), s2(s.size());
    for (auto &v : s) v = (int32_t) rng();
    std::vector<unsigned char> sb(5 * TFHE_HIP_IO_SAMPLE_BYTES);
    if (tfhe_hip_io_write_samples(sb.data(), sb.size(), &n, s.data(), nullptr, 5) || n != sb.size()) return 5;
    if (tfhe_hip_io_read_samples(sb.data(), n, s2.data(), nullptr, 5, &used) || s != s2) return 6;
    if (tfhe_hip_io_read_samples(sb.data(), n - 1, s2.data(), nullptr, 5, &used) == 0) return 7;
    if (tfhe_hip_load_cloud_keyset_buffer(nullptr, img.data(), 100) == 0) return 8;
    printf("ok: %d truncations rejected, %d of 3000 corrupted headers still parsed as the same parameter set\n", rejected, weird);
    return 0;
}
