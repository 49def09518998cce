// exa_build.cpp — turns a generated HIP module (exa_gen_*.cpp) into a gfx950 code object.
//
// The reference specialises its kernels inside the process at first call (Julia's JIT behind
// ext/ExaModelsKernelAbstractions.jl:608-653).  Here:
//   1. code objects handed over in memory (exa_cache_add: packed libraries, ExaModelsCompiler's compile_library role);
//   2. the on-disk cache, keyed by SHA-256(source) and SHA-256(flags | compiler identity | arch);
//   3. IN-PROCESS compilation with hiprtc (the copy that sits next to the HIP runtime this process has loaded, so a
//      process hosting PyTorch uses PyTorch's ROCm and a Julia/C host the system one) — no hipcc on the consumer's box;
//   4. hipcc --genco as a subprocess (EXAHIP_COMPILER=hipcc, or when hiprtc cannot be loaded).
// Cache directory: $EXAHIP_CACHE_DIR, else <install>/kernel_cache when writable, else $XDG_CACHE_HOME/exahip or
// ~/.cache/exahip created 0700.  A directory is only used when it is a real directory owned by this user (or root) and
// not writable by group/others; files are created O_EXCL|O_NOFOLLOW under unpredictable names and renamed into place.
// Nothing is ever read from or written to a shared location such as /tmp.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <pwd.h>
#include <spawn.h>
#include <sys/wait.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <random>
#include <sstream>

#include "exa_build.hpp"

extern char **environ;

namespace exa {
namespace {

// ---- SHA-256 (FIPS 180-4) ------------------------------------------------------------------------------------
struct Sha256 {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    unsigned char buf[64];
    size_t fill = 0;
    uint64_t total = 0;
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const unsigned char *p) {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
            0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
            0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
            0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
            0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
            0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
            const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const void *data, size_t n) {
        const unsigned char *p = (const unsigned char *)data;
        total += n;
        while (n) {
            const size_t k = std::min(n, 64 - fill);
            memcpy(buf + fill, p, k);
            fill += k; p += k; n -= k;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    std::string hex() {
        const uint64_t bits = total * 8;
        const unsigned char one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        unsigned char len[8];
        for (int i = 0; i < 8; i++) len[i] = (unsigned char)(bits >> (56 - 8 * i));
        update(len, 8);
        char out[65];
        for (int i = 0; i < 8; i++) snprintf(out + 8 * i, 9, "%08x", h[i]);
        return out;
    }
};

std::string lib_dir() {
    Dl_info info;
    if (dladdr((void *)&sha256_hex, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        auto k = p.find_last_of('/');
        if (k != std::string::npos) return p.substr(0, k);
    }
    return ".";
}

// a directory this process may trust for code objects: a real directory (not a symlink), owned by this user or root,
// not writable by group or others
bool trusted_dir(const std::string &d) {
    struct stat st;
    if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
    if (st.st_uid != getuid() && st.st_uid != 0) return false;
    return (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
bool mkdir_p(const std::string &d, mode_t mode) {
    for (size_t k = 1; k <= d.size(); k++)
        if (k == d.size() || d[k] == '/') {
            const std::string sub = d.substr(0, k);
            if (mkdir(sub.c_str(), mode) != 0 && errno != EEXIST) return false;
        }
    return true;
}

std::string primary_dir() {
    const char *env = getenv("EXAHIP_CACHE_DIR");
    return env && *env ? std::string(env) : lib_dir() + "/../kernel_cache";
}
std::string user_dir() {
    const char *xdg = getenv("XDG_CACHE_HOME");
    if (xdg && *xdg == '/') return std::string(xdg) + "/exahip";
    const char *home = getenv("HOME");
    if (!(home && *home == '/')) {
        struct passwd *pw = getpwuid(getuid());
        home = pw ? pw->pw_dir : nullptr;
    }
    return home && *home == '/' ? std::string(home) + "/.cache/exahip" : std::string();
}

// directories that may hold cached modules (read), in search order
std::vector<std::string> read_dirs() {
    std::vector<std::string> v;
    for (const std::string &d : {primary_dir(), user_dir()})
        if (!d.empty() && trusted_dir(d)) v.push_back(d);
    return v;
}

std::atomic<uint64_t> g_tmp_counter{0};
std::string unique_suffix() {
    static const uint64_t seed = [] {
        std::random_device rd;
        return ((uint64_t)rd() << 32) ^ rd() ^ ((uint64_t)getpid() << 20) ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
    }();
    char b[64];
    snprintf(b, sizeof b, ".%d.%016llx.%llu.tmp", (int)getpid(), (unsigned long long)seed, (unsigned long long)g_tmp_counter++);
    return b;
}
// writes `data` to a fresh file next to `final_path` (O_EXCL | O_NOFOLLOW, 0600) and renames it into place
void write_atomically(const std::string &final_path, const void *data, size_t n) {
    const std::string tmp = final_path + unique_suffix();
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) throw std::runtime_error("cannot create " + tmp + ": " + strerror(errno));
    const char *p = (const char *)data;
    size_t left = n;
    while (left) {
        const ssize_t k = write(fd, p, left);
        if (k < 0) { if (errno == EINTR) continue; const std::string e = strerror(errno); close(fd); unlink(tmp.c_str()); throw std::runtime_error("write " + tmp + ": " + e); }
        p += k; left -= (size_t)k;
    }
    if (fchmod(fd, 0644) != 0) { /* readable cache entries are a convenience only */ }
    close(fd);
    if (rename(tmp.c_str(), final_path.c_str()) != 0) {
        const std::string e = strerror(errno);
        unlink(tmp.c_str());
        throw std::runtime_error("rename " + tmp + " -> " + final_path + ": " + e);
    }
}

bool read_regular_file(const std::string &p, std::vector<char> &out) {
    const int fd = open(p.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0 || (st.st_uid != getuid() && st.st_uid != 0)) { close(fd); return false; }
    out.resize((size_t)st.st_size);
    size_t got = 0;
    while (got < out.size()) {
        const ssize_t k = read(fd, out.data() + got, out.size() - got);
        if (k <= 0) { if (k < 0 && errno == EINTR) continue; break; }
        got += (size_t)k;
    }
    close(fd);
    return got == out.size();
}

// an AMDGPU code object: ELF64 little-endian, e_machine = EM_AMDGPU (224)
bool looks_like_code_object(const void *blob, size_t len) {
    const unsigned char *p = (const unsigned char *)blob;
    if (len < 64 || memcmp(p, "\x7f" "ELF", 4) != 0 || p[4] != 2 || p[5] != 1) return false;
    return (p[18] | (p[19] << 8)) == 224;
}
// hipcc --genco wraps the ELF in a clang offload bundle; hipModuleLoadData accepts both
bool looks_like_bundle(const void *blob, size_t len) { return len > 24 && memcmp(blob, "__CLANG_OFFLOAD_BUNDLE__", 24) == 0; }

const char *kArch = "gfx950";
// THE GUARD AGAINST THE ONE WRONG-RESULT CLASS hipcc has shown here (tests/sweeps/canary/REPORT.md, profiles/NOTES.md round 5): the
// greedy SGPR allocator leaves live-range-split COPIES at the top of a control-flow join block, in front of the `s_or_b64 exec`
// that re-enables the lanes; SIInstrInfo::isBasicBlockPrologue does not count them as prologue, so everything the VGPR allocator
// later inserts "at the top of the block" (split copies, VGPR->AGPR copies, scratch spills, rematerialised constants) lands in front
// of the exec restore and runs under the narrowed mask: the other lanes keep whatever the register held.  Register counts do not
// show it (2 of the 6 affected kernels of round 4's cache had no AGPRs, no scratch, no spilled VGPRs).  The basic SGPR allocator
// does not split live ranges — it spills, and SGPR spills ARE prologue — which removes the cause for every kernel of every module,
// at no measurable cost (the kernels here are bound by VGPRs, not SGPRs; heavy kernels even spill fewer SGPRs:
// profiles/r5_sgpr_regalloc_ab.txt).  $EXAHIP_SGPR_REGALLOC = greedy restores the compiler's default (the canary's fault), fast /
// basic select explicitly.  tools/isa_prologue_check.py looks for the fault pattern in compiled code objects.
std::vector<std::string> base_flags() {
    std::vector<std::string> f = {std::string("--offload-arch=") + kArch, "-O3", "-std=c++17", "-munsafe-fp-atomics", "-w"};
    const char *ra = getenv("EXAHIP_SGPR_REGALLOC");
    const std::string alloc = ra && *ra ? ra : "basic";
    if (alloc != "greedy") { f.push_back("-mllvm"); f.push_back("-sgpr-regalloc=" + alloc); }
    const char *extra = getenv("EXAHIP_HIPCC_FLAGS");
    if (extra && *extra) {
        std::istringstream ss(extra);
        std::string t;
        while (ss >> t) f.push_back(t);
    }
    return f;
}
// Conservative register allocation for a module one of whose kernels has outgrown the 256 architectural VGPRs (AGPRs or
// scratch as spill space).  Such kernels have returned wrong, run-to-run different sums under the default allocator
// (profiles/NOTES.md round 3: exa_hprodw with 256 + 84 registers reads lanes it never wrote; tests/sweeps/canary); the same
// source is exact with region splitting of live ranges switched off.  $EXAHIP_SAFE_FLAGS overrides; "none" = no fallback.
std::vector<std::string> safe_flag_list() {
    const char *env = getenv("EXAHIP_SAFE_FLAGS");
    std::string text = env && *env ? env : "-mllvm -grow-region-complexity-budget=0";
    std::vector<std::string> f;
    if (text == "none") return f;
    std::istringstream ss(text);
    std::string t;
    while (ss >> t) f.push_back(t);
    return f;
}
std::string join(const std::vector<std::string> &v) {
    std::string s;
    for (const auto &t : v) { if (!s.empty()) s += ' '; s += t; }
    return s;
}

// ---- hiprtc, loaded from next to the HIP runtime of this process -----------------------------------------------
struct Rtc {
    void *lib = nullptr;
    std::string where, identity, why_not;
    hiprtcResult (*create)(hiprtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
    hiprtcResult (*compile)(hiprtcProgram, int, const char **) = nullptr;
    hiprtcResult (*log_size)(hiprtcProgram, size_t *) = nullptr;
    hiprtcResult (*log)(hiprtcProgram, char *) = nullptr;
    hiprtcResult (*code_size)(hiprtcProgram, size_t *) = nullptr;
    hiprtcResult (*code)(hiprtcProgram, char *) = nullptr;
    hiprtcResult (*destroy)(hiprtcProgram *) = nullptr;
    hiprtcResult (*version)(int *, int *) = nullptr;
};
Rtc &rtc() {
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::string> cands;
        Dl_info info;
        if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
            std::string p = info.dli_fname;
            auto k = p.find_last_of('/');
            if (k != std::string::npos) { cands.push_back(p.substr(0, k) + "/libhiprtc.so.7"); cands.push_back(p.substr(0, k) + "/libhiprtc.so"); }
        }
        cands.push_back("libhiprtc.so.7");
        cands.push_back("libhiprtc.so");
        for (const auto &c : cands) {
            r.lib = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (r.lib) { r.where = c; break; }
        }
        if (!r.lib) { r.why_not = std::string("libhiprtc could not be loaded: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char *n) { void *s = dlsym(r.lib, n); if (!s && r.why_not.empty()) r.why_not = std::string("libhiprtc lacks ") + n; return s; };
        r.create = (decltype(r.create))sym("hiprtcCreateProgram");
        r.compile = (decltype(r.compile))sym("hiprtcCompileProgram");
        r.log_size = (decltype(r.log_size))sym("hiprtcGetProgramLogSize");
        r.log = (decltype(r.log))sym("hiprtcGetProgramLog");
        r.code_size = (decltype(r.code_size))sym("hiprtcGetCodeSize");
        r.code = (decltype(r.code))sym("hiprtcGetCode");
        r.destroy = (decltype(r.destroy))sym("hiprtcDestroyProgram");
        r.version = (decltype(r.version))sym("hiprtcVersion");
        if (!r.why_not.empty()) { dlclose(r.lib); r.lib = nullptr; return; }
        int mj = 0, mn = 0, rt = 0;
        r.version(&mj, &mn);
        (void)hipRuntimeGetVersion(&rt);
        r.identity = "hiprtc-" + std::to_string(mj) + "." + std::to_string(mn) + "-hip" + std::to_string(rt);
    });
    return r;
}

std::string hipcc_path() {
    const char *cc = getenv("EXAHIP_HIPCC");
    return cc && *cc ? cc : "/opt/rocm/bin/hipcc";
}
// first line of `hipcc --version` (once per path and process)
std::string hipcc_identity(const std::string &path) {
    static std::mutex mu;
    static std::map<std::string, std::string> memo;
    std::lock_guard<std::mutex> lk(mu);
    auto it = memo.find(path);
    if (it != memo.end()) return it->second;
    std::string id = "hipcc:" + path;
    if (access(path.c_str(), X_OK) == 0) {
        const std::string cmd = "'" + path + "' --version 2>/dev/null";
        if (FILE *f = popen(cmd.c_str(), "r")) {
            char line[512];
            for (int k = 0; k < 4 && fgets(line, sizeof line, f); k++) id += std::string("|") + line;
            pclose(f);
        }
    }
    return memo[path] = id;
}

enum class Tool { Hiprtc, Hipcc };
Tool pick_tool(std::string &identity) {
    const char *env = getenv("EXAHIP_COMPILER");
    const std::string want = env ? env : "auto";
    if (want != "auto" && want != "hiprtc" && want != "hipcc") throw std::runtime_error("EXAHIP_COMPILER must be auto, hiprtc or hipcc");
    if (want != "hipcc") {
        Rtc &r = rtc();
        if (r.lib) { identity = r.identity; return Tool::Hiprtc; }
        if (want == "hiprtc") throw std::runtime_error("EXAHIP_COMPILER=hiprtc: " + r.why_not);
    }
    identity = hipcc_identity(hipcc_path());
    return Tool::Hipcc;
}

// hiprtc in a process of its own (csrc/exa_rtc_helper.cpp: LLVM latches -sgpr-regalloc at the first compilation of a process, so
// inside a host that has compiled through this comgr before, the guard flag of base_flags() would be without effect).  The helper
// sits next to libexahip.so; without it (a partial install), or under EXAHIP_RTC_INPROCESS=1, the compilation runs in this process
// — correct whenever this library's compilations are the process's first through that comgr, which is the normal case.
// Returns false when the helper is not there; throws on a compile error, like the in-process path.
bool compile_hiprtc_helper(const std::string &source, const std::vector<std::string> &flags, std::vector<char> &image) {
    const std::string helper = lib_dir() + "/exa_rtc";
    if (access(helper.c_str(), X_OK) != 0) return false;
    Rtc &r = rtc();
    // a private directory of this user (0700, mkdtemp): source in, code object and log out
    const char *td = getenv("TMPDIR");
    std::string tmpl = std::string(td && *td == '/' ? td : "/tmp") + "/exa_rtc.XXXXXX";
    std::vector<char> dirbuf(tmpl.begin(), tmpl.end());
    dirbuf.push_back('\0');
    if (!mkdtemp(dirbuf.data())) return false;
    const std::string dir = dirbuf.data(), src = dir + "/m.hip", obj = dir + "/m.hsaco", log = dir + "/m.log";
    auto cleanup = [&] { unlink(src.c_str()); unlink(obj.c_str()); unlink(log.c_str()); rmdir(dir.c_str()); };
    try {
        { std::ofstream f(src, std::ios::binary); f.write(source.data(), (std::streamsize)source.size()); if (!f) throw std::runtime_error("cannot write " + src); }
        std::vector<std::string> args = {helper, r.where, src, obj};
        for (const auto &f : flags) args.push_back(f);
        std::vector<char *> argv;
        for (auto &a : args) argv.push_back(const_cast<char *>(a.c_str()));
        argv.push_back(nullptr);
        posix_spawn_file_actions_t fa;
        posix_spawn_file_actions_init(&fa);
        posix_spawn_file_actions_addopen(&fa, 2, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
        posix_spawn_file_actions_addopen(&fa, 1, "/dev/null", O_WRONLY, 0);
        pid_t pid = 0;
        const int rc = posix_spawn(&pid, helper.c_str(), &fa, nullptr, argv.data(), environ);
        posix_spawn_file_actions_destroy(&fa);
        if (rc != 0) { cleanup(); return false; }
        int status = 0;
        while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
        const int code = WIFEXITED(status) ? WEXITSTATUS(status) : 2;
        if (code == 0 && read_regular_file(obj, image) && looks_like_code_object(image.data(), image.size())) { cleanup(); return true; }
        std::vector<char> l;
        std::string msg;
        if (read_regular_file(log, l)) msg.assign(l.begin(), l.end());
        if (msg.size() > 4000) msg.resize(4000);
        cleanup();
        if (code == 1) throw std::runtime_error("hiprtc failed (" + r.where + " in exa_rtc, " + join(flags) + "):\n" + msg);
        if (WIFSIGNALED(status)) throw std::runtime_error("hiprtc failed: the compiler process died with signal " + std::to_string(WTERMSIG(status)) + "\n" + msg);
        return false;       // the helper could not run (its libhiprtc did not load, ...): the in-process path will say why
    } catch (...) { cleanup(); throw; }
}

// Explicit opt-in to hiprtc INSIDE the host process (EXAHIP_RTC_INPROCESS=1).  Never a silent fallback (round 6): a host that compiled
// anything through this comgr before us — a Julia process with AMDGPU.jl does — has latched the greedy SGPR allocator, and -sgpr-regalloc
// in base_flags() is then silently ignored.
bool rtc_inprocess_requested() {
    const char *inproc = getenv("EXAHIP_RTC_INPROCESS");
    return inproc && *inproc && std::string(inproc) != "0";
}

std::vector<char> compile_hiprtc(const std::string &source, const std::vector<std::string> &flags) {
    if (!rtc_inprocess_requested()) {
        std::vector<char> image;
        if (compile_hiprtc_helper(source, flags, image)) return image;
        // no helper (partial install), or it could not run: REFUSE — the guard flag means nothing in a process whose LLVM may have latched
        // another allocator, and a wrong-result code object would then be trusted by every later process through the disk cache
        throw std::runtime_error("the kernel compiler process " + lib_dir() + "/exa_rtc is missing or could not run: refusing to compile inside the host process "
                                 "(LLVM latches -sgpr-regalloc at a process's first compilation; a host that compiled through this comgr before would disarm "
                                 "the guard against the known miscompilation, tests/sweeps/canary/REPORT.md).  Install exa_rtc next to libexahip.so, set "
                                 "EXAHIP_COMPILER=hipcc, or opt in with EXAHIP_RTC_INPROCESS=1 (compiled with the conservative allocator flags as well, cached under another name)");
    }
    static std::once_flag warned;
    std::call_once(warned, [] {
        fprintf(stderr, "[exahip] WARNING: EXAHIP_RTC_INPROCESS=1: kernels are compiled by hiprtc inside this process; -sgpr-regalloc=basic only takes effect if no "
                        "compilation ran in this process before.  Modules are built with the conservative allocator flags in addition and cached under a name of their own.\n");
    });
    Rtc &r = rtc();
    hiprtcProgram prog = nullptr;
    if (r.create(&prog, source.c_str(), "exa_module.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) throw std::runtime_error("hiprtcCreateProgram failed");
    std::vector<const char *> opts;
    for (const auto &f : flags) opts.push_back(f.c_str());
    const hiprtcResult rc = r.compile(prog, (int)opts.size(), opts.data());
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        r.log_size(prog, &n);
        std::string log(n + 1, '\0');
        if (n) r.log(prog, log.data());
        r.destroy(&prog);
        if (log.size() > 4000) log.resize(4000);
        throw std::runtime_error("hiprtc failed (" + r.where + ", " + join(flags) + "):\n" + log.c_str());
    }
    size_t n = 0;
    r.code_size(prog, &n);
    std::vector<char> image(n);
    if (n) r.code(prog, image.data());
    r.destroy(&prog);
    if (!looks_like_code_object(image.data(), image.size())) throw std::runtime_error("hiprtc returned something that is not a gfx950 code object");
    return image;
}

std::vector<char> compile_hipcc(const std::string &source, const std::string &workdir, const std::vector<std::string> &flags) {
    if (workdir.empty()) throw std::runtime_error("hipcc needs a writable cache directory (set EXAHIP_CACHE_DIR) and none was found");
    const std::string stem = workdir + "/build" + unique_suffix();
    const std::string src = stem + ".hip", obj = stem + ".hsaco", log = stem + ".log";
    write_atomically(src, source.data(), source.size());
    auto q = [](const std::string &p) { return "'" + p + "'"; };   // paths may contain spaces; they never contain quotes
    const std::string cmd = q(hipcc_path()) + " --genco " + join(flags) + " -o " + q(obj) + " " + q(src) + " > " + q(log) + " 2>&1";
    const int rc = std::system(cmd.c_str());
    std::vector<char> image;
    const bool ok = rc == 0 && read_regular_file(obj, image);
    std::string msg;
    if (!ok) {
        std::vector<char> l;
        if (read_regular_file(log, l)) msg.assign(l.begin(), l.end());
        if (msg.size() > 4000) msg.resize(4000);
    }
    unlink(src.c_str()); unlink(obj.c_str()); unlink(log.c_str());
    if (!ok) throw std::runtime_error("hipcc failed (" + cmd + "):\n" + msg);
    return image;
}

std::mutex g_pre_mu;
std::map<std::string, std::vector<char>> g_preloaded;      // source key -> code object

}  // namespace

std::string sha256_hex(const std::string &s) {
    Sha256 h;
    h.update(s.data(), s.size());
    return h.hex();
}

std::string source_key(const std::string &source) { return "exa_" + sha256_hex(source).substr(0, 32); }

std::string writable_cache_dir() {
    const std::string p = primary_dir();
    const bool from_env = getenv("EXAHIP_CACHE_DIR") && *getenv("EXAHIP_CACHE_DIR");
    if (!trusted_dir(p)) { if (mkdir_p(p, from_env ? 0700 : 0755)) { /* created */ } }
    if (trusted_dir(p) && access(p.c_str(), W_OK) == 0) return p;
    const std::string u = user_dir();
    if (!u.empty()) {
        if (!trusted_dir(u)) mkdir_p(u, 0700);
        if (trusted_dir(u) && access(u.c_str(), W_OK) == 0) return u;
    }
    return std::string();
}

bool cache_add(const std::string &name, const void *blob, size_t len) {
    if (name.compare(0, 4, "exa_") != 0 || !blob) return false;
    if (!looks_like_code_object(blob, len) && !looks_like_bundle(blob, len)) return false;
    std::lock_guard<std::mutex> lk(g_pre_mu);
    g_preloaded[name] = std::vector<char>((const char *)blob, (const char *)blob + len);
    return true;
}

std::string safe_flags() { return join(safe_flag_list()); }
bool cache_has(const std::string &name) {
    std::lock_guard<std::mutex> lk(g_pre_mu);
    return g_preloaded.count(name) != 0;
}

CodeObject get_code_object(const std::string &source, bool memory_only_ok, bool safe) {
    CodeObject co;
    co.key = source_key(source);
    co.safe = safe;
    co.name = co.key + (safe ? "_safe" : "");
    const auto t0 = std::chrono::steady_clock::now();
    if (memory_only_ok) {      // a caller that wants the FILE (exa_compile: packing) gets a real cache entry instead
        std::lock_guard<std::mutex> lk(g_pre_mu);
        auto it = g_preloaded.find(co.name);
        if (it != g_preloaded.end()) { co.image = it->second; co.how = "preloaded"; co.path = "(preloaded) " + co.name; return co; }
    }
    std::string identity;
    const Tool tool = pick_tool(identity);
    std::vector<std::string> flags = base_flags();
    // in-process hiprtc (explicit opt-in): the allocator flag may be latched away, so the region-splitting budget — an ordinary option, read at
    // every compilation, which removes the fault site on its own (profiles/r5_guard_sweeps.txt) — always rides along, and the cache name says so
    const bool inproc = tool == Tool::Hiprtc && rtc_inprocess_requested();
    if (safe || inproc) for (const std::string &f : safe_flag_list()) flags.push_back(f);
    if (inproc) identity += "|inprocess";
    const std::string file = co.key + "-" + sha256_hex(join(flags) + "|" + identity + "|" + kArch).substr(0, 12) + ".hsaco";
    for (const std::string &d : read_dirs()) {
        const std::string p = d + "/" + file;
        if (read_regular_file(p, co.image) && (looks_like_code_object(co.image.data(), co.image.size()) || looks_like_bundle(co.image.data(), co.image.size()))) {
            co.how = "disk"; co.path = p;
            (void)utimensat(AT_FDCWD, p.c_str(), nullptr, 0);      // "last used" stamp: what a cache pruner goes by (a read-only cache just keeps its dates)
            return co;
        }
    }
    const std::string wdir = writable_cache_dir();
    if (tool == Tool::Hiprtc) { co.image = compile_hiprtc(source, flags); co.how = inproc ? "hiprtc-inprocess" : "hiprtc"; }
    else { co.image = compile_hipcc(source, wdir, flags); co.how = "hipcc"; }
    co.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (!wdir.empty()) {
        co.path = wdir + "/" + file;
        try {
            write_atomically(co.path, co.image.data(), co.image.size());
            if (getenv("EXAHIP_KEEP_SOURCE")) write_atomically(wdir + "/" + co.key + ".hip", source.data(), source.size());
        } catch (const std::exception &) {
            if (!memory_only_ok) throw;
            co.path = "(memory) " + co.name;      // a cache that cannot be written is not an error for a loaded module
        }
    } else {
        if (!memory_only_ok) throw std::runtime_error("no writable kernel cache directory (EXAHIP_CACHE_DIR, <install>/kernel_cache, ~/.cache/exahip)");
        co.path = "(memory) " + co.name;
    }
    return co;
}

// ---- notes about a module: <cache>/<source key>.note, or handed over in memory (exa_cache_note) ----------------------------
// One fact so far: "loopfree" — the scatter kernels of this module spill registers when compiled with loops around their
// bodies, so the model's module is the one generated WITHOUT them (another source, another key).  The decision is taken
// once, where the module is compiled (exa_compile, exahip.pack, or the first device build), from the code object's own
// metadata; every later build — plan-only or device, this process or another, a packed library's consumer — starts from
// the note and generates the final module at once: no second compile, and no compiler needed by a consumer.
std::map<std::string, std::string> g_notes;
std::string note_lookup(const std::string &key) {
    {
        std::lock_guard<std::mutex> lk(g_pre_mu);
        auto it = g_notes.find(key);
        if (it != g_notes.end()) return it->second;
    }
    for (const std::string &d : read_dirs()) {
        std::vector<char> txt;
        if (read_regular_file(d + "/" + key + ".note", txt)) {
            std::string t(txt.begin(), txt.end());
            while (!t.empty() && (t.back() == '\n' || t.back() == ' ')) t.pop_back();
            return t;
        }
    }
    return "";
}
void note_store(const std::string &key, const std::string &note, bool persist) {
    { std::lock_guard<std::mutex> lk(g_pre_mu); g_notes[key] = note; }
    if (!persist) return;
    const std::string d = writable_cache_dir();
    if (d.empty()) return;
    try { write_atomically(d + "/" + key + ".note", note.data(), note.size()); } catch (const std::exception &) {}
}

// ---- resources of the kernels, read from the code object's AMDGPU metadata (msgpack in an ELF note) -------------------------
// ELF64 -> SHT_NOTE sections (or PT_NOTE segments) -> the note (name "AMDGPU", type NT_AMDGPU_METADATA = 32) -> msgpack map
// { "amdhsa.kernels": [ { ".name": ..., ".vgpr_count": ..., ".agpr_count": ..., ".private_segment_fixed_size": ..., ... } ] }.
// A real (if minimal) msgpack reader: every object is either understood or skipped by its encoded length, so a kernel's
// numbers can only come from that kernel's own map.  hipcc --genco wraps the ELF in a clang offload bundle: its amdgcn entry
// is unwrapped first.
namespace {
struct MsgPack {
    const unsigned char *p, *end;
    bool ok = true;
    bool need(size_t n) { if ((size_t)(end - p) < n) { ok = false; return false; } return true; }
    uint64_t be(int n) { uint64_t v = 0; for (int i = 0; i < n; i++) v = (v << 8) | p[i]; p += n; return v; }
    // header of the next object: kind 'm' map, 'a' array, 's' str, 'b' opaque bytes (len of them follow), 'u' unsigned,
    // 'i' signed, 'n' nil / bool; len = element count / byte length / value
    bool head(char &kind, uint64_t &len) {
        if (!need(1)) return false;
        const unsigned char c = *p++;
        if (c <= 0x7f) { kind = 'u'; len = c; return true; }
        if (c <= 0x8f) { kind = 'm'; len = c & 0x0f; return true; }
        if (c <= 0x9f) { kind = 'a'; len = c & 0x0f; return true; }
        if (c <= 0xbf) { kind = 's'; len = c & 0x1f; return true; }
        if (c >= 0xe0) { kind = 'i'; len = (uint64_t)(int64_t)(signed char)c; return true; }
        auto sized = [&](char k, int n) { if (!need((size_t)n)) return false; kind = k; len = be(n); return true; };
        switch (c) {
        case 0xc0: case 0xc2: case 0xc3: kind = 'n'; len = c == 0xc3; return true;
        case 0xc4: return sized('b', 1);
        case 0xc5: return sized('b', 2);
        case 0xc6: return sized('b', 4);
        case 0xc7: if (!sized('b', 1)) return false; len += 1; return true;     // ext: + its type byte
        case 0xc8: if (!sized('b', 2)) return false; len += 1; return true;
        case 0xc9: if (!sized('b', 4)) return false; len += 1; return true;
        case 0xca: kind = 'b'; len = 4; return true;                            // float32 / float64: skipped as bytes
        case 0xcb: kind = 'b'; len = 8; return true;
        case 0xcc: return sized('u', 1);
        case 0xcd: return sized('u', 2);
        case 0xce: return sized('u', 4);
        case 0xcf: return sized('u', 8);
        case 0xd0: if (!sized('i', 1)) return false; len = (uint64_t)(int64_t)(int8_t)len; return true;
        case 0xd1: if (!sized('i', 2)) return false; len = (uint64_t)(int64_t)(int16_t)len; return true;
        case 0xd2: if (!sized('i', 4)) return false; len = (uint64_t)(int64_t)(int32_t)len; return true;
        case 0xd3: return sized('i', 8);
        case 0xd4: kind = 'b'; len = 2; return true;                            // fixext 1 / 2 / 4 / 8 / 16
        case 0xd5: kind = 'b'; len = 3; return true;
        case 0xd6: kind = 'b'; len = 5; return true;
        case 0xd7: kind = 'b'; len = 9; return true;
        case 0xd8: kind = 'b'; len = 17; return true;
        case 0xd9: return sized('s', 1);
        case 0xda: return sized('s', 2);
        case 0xdb: return sized('s', 4);
        case 0xdc: return sized('a', 2);
        case 0xdd: return sized('a', 4);
        case 0xde: return sized('m', 2);
        case 0xdf: return sized('m', 4);
        }
        ok = false;
        return false;
    }
    bool skip(int depth = 0) {
        char k; uint64_t n;
        if (depth > 64 || !head(k, n)) { ok = false; return false; }
        if (k == 's' || k == 'b') { if (!need((size_t)n)) return false; p += n; return true; }
        if (k == 'a') { for (uint64_t i = 0; i < n; i++) if (!skip(depth + 1)) return false; return true; }
        if (k == 'm') { for (uint64_t i = 0; i < 2 * n; i++) if (!skip(depth + 1)) return false; return true; }
        return true;
    }
    bool str(std::string &out) {
        char k; uint64_t n;
        if (!head(k, n) || k != 's' || !need((size_t)n)) { ok = false; return false; }
        out.assign((const char *)p, (size_t)n); p += n;
        return true;
    }
};
uint64_t le(const unsigned char *q, int n) { uint64_t v = 0; for (int i = n - 1; i >= 0; i--) v = (v << 8) | q[i]; return v; }
// the gfx950 ELF inside a clang offload bundle ("__CLANG_OFFLOAD_BUNDLE__", u64 entries, then per entry offset, size, triple)
bool unbundle(const unsigned char *&b, size_t &n) {
    if (n < 32 || memcmp(b, "__CLANG_OFFLOAD_BUNDLE__", 24) != 0) return true;      // not a bundle: as it is
    const uint64_t cnt = le(b + 24, 8);
    size_t at = 32;
    for (uint64_t e = 0; e < cnt && e < 64; e++) {
        if (at + 24 > n) return false;
        const uint64_t off = le(b + at, 8), size = le(b + at + 8, 8), tl = le(b + at + 16, 8);
        at += 24;
        if (tl > n || at + tl > n) return false;
        const std::string triple((const char *)b + at, (size_t)tl);
        at += (size_t)tl;
        if (triple.find("amdgcn") != std::string::npos && off <= n && size <= n - off && size >= 64) { b += off; n = (size_t)size; return true; }
    }
    return false;
}
bool parse_metadata(const unsigned char *d, size_t n, std::vector<KernelInfo> &out) {
    MsgPack mp{d, d + n};
    char k; uint64_t top;
    if (!mp.head(k, top) || k != 'm') return false;
    bool found = false;
    for (uint64_t i = 0; i < top && mp.ok; i++) {
        std::string key;
        if (!mp.str(key)) return false;
        if (key != "amdhsa.kernels") { if (!mp.skip()) return false; continue; }
        uint64_t nk;
        if (!mp.head(k, nk) || k != 'a') return false;
        found = true;
        for (uint64_t j = 0; j < nk; j++) {
            uint64_t nf;
            if (!mp.head(k, nf) || k != 'm') return false;
            KernelInfo ki;
            bool have_v = false, have_sc = false;
            for (uint64_t f = 0; f < nf; f++) {
                std::string fk;
                if (!mp.str(fk)) return false;
                if (fk == ".name") { if (!mp.str(ki.name)) return false; continue; }
                int *dst = fk == ".vgpr_count" ? &ki.vgpr : fk == ".agpr_count" ? &ki.agpr : fk == ".sgpr_count" ? &ki.sgpr : fk == ".private_segment_fixed_size" ? &ki.scratch
                         : fk == ".vgpr_spill_count" ? &ki.vgpr_spill : fk == ".sgpr_spill_count" ? &ki.sgpr_spill : fk == ".group_segment_fixed_size" ? &ki.lds : nullptr;
                if (!dst) { if (!mp.skip()) return false; continue; }
                uint64_t v;
                if (!mp.head(k, v) || (k != 'u' && k != 'i')) return false;
                *dst = (int)v;
                have_v = have_v || fk == ".vgpr_count";
                have_sc = have_sc || fk == ".private_segment_fixed_size";
            }
            if (ki.name.empty() || !have_v || !have_sc) return false;      // a kernel entry without its basic facts: not understood
            out.push_back(ki);
        }
    }
    return found && mp.ok;
}
}  // namespace

bool code_object_kernels(const std::vector<char> &image, std::vector<KernelInfo> &out) {
    out.clear();
    const unsigned char *b = (const unsigned char *)image.data();
    size_t n = image.size();
    if (!unbundle(b, n) || !looks_like_code_object(b, n)) return false;
    // notes: SHT_NOTE sections, else PT_NOTE segments
    std::vector<std::pair<uint64_t, uint64_t>> notes;
    const uint64_t shoff = le(b + 0x28, 8), phoff = le(b + 0x20, 8);
    const unsigned shentsize = (unsigned)le(b + 0x3a, 2), shnum = (unsigned)le(b + 0x3c, 2), phentsize = (unsigned)le(b + 0x36, 2), phnum = (unsigned)le(b + 0x38, 2);
    if (shoff && shentsize >= 64 && shoff <= n && (uint64_t)shentsize * shnum <= n - shoff)
        for (unsigned i = 0; i < shnum; i++) {
            const unsigned char *sh = b + shoff + (uint64_t)i * shentsize;
            if (le(sh + 4, 4) == 7) notes.push_back({le(sh + 0x18, 8), le(sh + 0x20, 8)});
        }
    if (notes.empty() && phoff && phentsize >= 56 && phoff <= n && (uint64_t)phentsize * phnum <= n - phoff)
        for (unsigned i = 0; i < phnum; i++) {
            const unsigned char *ph = b + phoff + (uint64_t)i * phentsize;
            if (le(ph, 4) == 4) notes.push_back({le(ph + 8, 8), le(ph + 0x20, 8)});
        }
    for (const auto &nt : notes) {
        if (nt.first > n || nt.second > n - nt.first) continue;
        uint64_t at = nt.first;
        const uint64_t stop = nt.first + nt.second;
        while (at + 12 <= stop) {
            const uint64_t namesz = le(b + at, 4), descsz = le(b + at + 4, 4), type = le(b + at + 8, 4);
            const uint64_t name_at = at + 12, desc_at = name_at + ((namesz + 3) & ~(uint64_t)3);
            if (desc_at > stop || descsz > stop - desc_at) break;
            if (type == 32 && namesz >= 6 && memcmp(b + name_at, "AMDGPU", 6) == 0) return parse_metadata(b + desc_at, (size_t)descsz, out);
            at = desc_at + ((descsz + 3) & ~(uint64_t)3);
        }
    }
    return false;
}

bool kernel_resources(const std::vector<char> &image, const std::string &kernel, int *vgpr, int *agpr, int *scratch, int *vgpr_spill, int *sgpr_spill) {
    std::vector<KernelInfo> ks;
    if (!code_object_kernels(image, ks)) return false;
    for (const KernelInfo &k : ks)
        if (k.name == kernel) {
            *vgpr = k.vgpr; *agpr = k.agpr; *scratch = k.scratch; *vgpr_spill = k.vgpr_spill;
            if (sgpr_spill) *sgpr_spill = k.sgpr_spill;
            return true;
        }
    return false;
}

// ---- persisted tuning decisions: <cache>/<source key>.tune, lines "<signature> <value>" ---------------------------
static bool tune_lookup_in(const std::string &d, const std::string &key, const std::string &signature, int *value) {
    std::vector<char> txt;
    if (!read_regular_file(d + "/" + key + ".tune", txt)) return false;
    std::istringstream ss(std::string(txt.begin(), txt.end()));
    std::string sig;
    int v, found = 0;
    while (ss >> sig >> v) if (sig == signature) { *value = v; found = 1; }    // the last line for a signature wins
    return found != 0;
}
// The directory decisions are WRITTEN to is read first: a newer exa_tune result there is not shadowed by an older line in
// a read-only <install>/kernel_cache that comes earlier in read_dirs().
bool tune_lookup(const std::string &key, const std::string &signature, int *value) {
    const std::string w = writable_cache_dir();
    if (!w.empty() && tune_lookup_in(w, key, signature, value)) return true;
    for (const std::string &d : read_dirs())
        if (d != w && tune_lookup_in(d, key, signature, value)) return true;
    return false;
}
void tune_store(const std::string &key, const std::string &signature, int value) {
    const std::string d = writable_cache_dir();
    if (d.empty()) return;
    const std::string p = d + "/" + key + ".tune";
    // one O_APPEND write per decision: the ranks of a multi-GPU job store their (rank-specific) lines into the same file
    // at the same moment, and a read-modify-write would keep only the last writer's.  The directory is a trusted one
    // (writable_cache_dir); the last line for a signature wins when the file is read.
    int cur = 0;
    if (tune_lookup_in(d, key, signature, &cur) && cur == value) return;      // nothing new IN THE FILE BEING WRITTEN: it does not grow with every exa_tune
    const std::string line = signature + " " + std::to_string(value) + "\n";
    const int fd = open(p.c_str(), O_WRONLY | O_CREAT | O_APPEND | O_NOFOLLOW | O_CLOEXEC, 0644);
    if (fd < 0) return;                                     // tuning is an optimisation
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == getuid()) { const ssize_t k = write(fd, line.data(), line.size()); (void)k; }
    close(fd);
}

}  // namespace exa
