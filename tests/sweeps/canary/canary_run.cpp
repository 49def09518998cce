// canary_run.cpp — runs ONE recorded kernel launch (EXADUMP1, see canary_host.cpp) against a code object ONCE and writes the raw
// output vector to a file.  TEST INFRASTRUCTURE for scan_uninit.py (which registers / lanes / instructions of the default build
// read a register lane the kernel never wrote) — nothing in the product uses it.
//   hipcc -O2 -o canary_run canary_run.cpp;  ./canary_run DUMP CODE_OBJECT OUT.bin     exit 0 = ran, 2 = could not run
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 2; } } while (0)

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: canary_run DUMP CODE_OBJECT OUT.bin\n"); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    if (!f) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    auto word = [&] { long long q = 0; f.read((char *)&q, 8); return q; };
    char magic[8];
    f.read(magic, 8);
    if (memcmp(magic, "EXADUMP1", 8) != 0) { fprintf(stderr, "not an EXADUMP1 file\n"); return 2; }
    std::string kname((size_t)word(), '\0');
    f.read(kname.data(), (std::streamsize)kname.size());
    const long long grid = word(), block = word(), lds = word(), nargs = word();
    std::vector<std::vector<char>> scalars((size_t)nargs);
    std::vector<void *> dev((size_t)nargs, nullptr), args((size_t)nargs, nullptr);
    long long out_arg = -1, out_bytes = 0;
    for (long long a = 0; a < nargs; a++) {
        const long long kind = word(), n = word();
        if (kind == 0) { scalars[a].resize((size_t)n); f.read(scalars[a].data(), n); args[a] = scalars[a].data(); continue; }
        CHK(hipMalloc(&dev[a], (size_t)(n ? n : 8)));
        if (kind == 1) { std::vector<char> tmp((size_t)n); f.read(tmp.data(), n); CHK(hipMemcpy(dev[a], tmp.data(), (size_t)n, hipMemcpyHostToDevice)); }
        else { out_arg = a; out_bytes = n; }
        args[a] = &dev[a];
    }
    const long long nexp = word();
    std::vector<double> expect((size_t)nexp), got((size_t)nexp);
    f.read((char *)expect.data(), 8 * nexp);
    if (!f || out_arg < 0 || out_bytes != 8 * nexp) { fprintf(stderr, "truncated or malformed dump\n"); return 2; }
    std::ifstream cf(argv[2], std::ios::binary);
    std::vector<char> image((std::istreambuf_iterator<char>(cf)), std::istreambuf_iterator<char>());
    if (image.empty()) { fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
    hipModule_t mod;
    hipFunction_t fn;
    CHK(hipModuleLoadData(&mod, image.data()));
    CHK(hipModuleGetFunction(&fn, mod, kname.c_str()));
    std::vector<double> nan((size_t)nexp, NAN);
    CHK(hipMemcpy(dev[out_arg], nan.data(), (size_t)out_bytes, hipMemcpyHostToDevice));
    CHK(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, (unsigned)block, 1, 1, (unsigned)lds, nullptr, args.data(), nullptr));
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(got.data(), dev[out_arg], (size_t)out_bytes, hipMemcpyDeviceToHost));
    std::ofstream o(argv[3], std::ios::binary);
    o.write((const char *)expect.data(), 8 * nexp);        // first half: the recorded (expected) output, second half: this run's
    o.write((const char *)got.data(), 8 * nexp);
    return o ? 0 : 2;
}
