// exa_rtc — libexahip's kernel compiler as a PROCESS OF ITS OWN: loads libhiprtc, compiles one HIP source for gfx950, writes the
// code object.  Why not inside the host process: LLVM's option state is per process and partly LATCHED — the AMDGPU back end picks
// its SGPR register allocator (-sgpr-regalloc) once, at the first compilation of the process, and ignores the option afterwards.
// libexahip's guard against the compiler fault of tests/sweeps/canary/REPORT.md IS that option; inside a host that has already
// compiled something through the same comgr (PyTorch's jiterator, MIOpen, another library) it would be silently without effect.  A
// fresh process per compilation makes every flag mean what it says (and keeps a compiler crash out of the solver's process).
//   exa_rtc LIBHIPRTC SOURCE.hip OUT.hsaco [flags...]        exit 0 = written; 1 = compile error (log on stderr); 2 = could not run
#include <dlfcn.h>

#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

typedef struct _hiprtcProgram *hiprtcProgram;
typedef int hiprtcResult;      // HIPRTC_SUCCESS = 0

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: exa_rtc LIBHIPRTC SOURCE OUT [flags...]\n"); return 2; }
    void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "exa_rtc: %s\n", dlerror()); return 2; }
    auto create = (hiprtcResult (*)(hiprtcProgram *, const char *, const char *, int, const char **, const char **))dlsym(lib, "hiprtcCreateProgram");
    auto compile = (hiprtcResult (*)(hiprtcProgram, int, const char **))dlsym(lib, "hiprtcCompileProgram");
    auto log_size = (hiprtcResult (*)(hiprtcProgram, size_t *))dlsym(lib, "hiprtcGetProgramLogSize");
    auto log = (hiprtcResult (*)(hiprtcProgram, char *))dlsym(lib, "hiprtcGetProgramLog");
    auto code_size = (hiprtcResult (*)(hiprtcProgram, size_t *))dlsym(lib, "hiprtcGetCodeSize");
    auto code = (hiprtcResult (*)(hiprtcProgram, char *))dlsym(lib, "hiprtcGetCode");
    if (!create || !compile || !log_size || !log || !code_size || !code) { fprintf(stderr, "exa_rtc: %s lacks the hiprtc entry points\n", argv[1]); return 2; }
    std::ifstream f(argv[2], std::ios::binary);
    if (!f) { fprintf(stderr, "exa_rtc: cannot read %s\n", argv[2]); return 2; }
    const std::string source((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    hiprtcProgram prog = nullptr;
    if (create(&prog, source.c_str(), "exa_module.hip", 0, nullptr, nullptr) != 0) { fprintf(stderr, "exa_rtc: hiprtcCreateProgram failed\n"); return 2; }
    std::vector<const char *> opts(argv + 4, argv + argc);
    if (compile(prog, (int)opts.size(), opts.data()) != 0) {
        size_t n = 0;
        log_size(prog, &n);
        std::string text(n + 1, '\0');
        if (n) log(prog, &text[0]);
        fputs(text.c_str(), stderr);
        return 1;
    }
    size_t n = 0;
    code_size(prog, &n);
    std::vector<char> image(n);
    if (n) code(prog, image.data());
    std::ofstream o(argv[3], std::ios::binary);
    o.write(image.data(), (std::streamsize)image.size());
    o.close();
    return o ? 0 : 2;
}
