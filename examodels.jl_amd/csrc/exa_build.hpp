// exa_build.hpp — code-object build + cache + persisted tuning decisions (exa_build.cpp)
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

namespace exa {

std::string sha256_hex(const std::string &s);
// "exa_" + first 128 bits of SHA-256(source): names the generated module independently of the compiler (exa_module_name,
// exa_cache_add)
std::string source_key(const std::string &source);

struct CodeObject {
    std::string key;            // source_key
    std::vector<char> image;    // gfx950 code object (ELF) or clang offload bundle
    std::string path;           // cache file, or "(preloaded) key" / "(memory) key"
    std::string how;            // preloaded | disk | hiprtc | hipcc
    double build_ms = 0.0;      // compiler time when it ran
    bool safe = false;          // compiled with the conservative register-allocation flags (safe_flags)
    std::string name;           // what exa_cache_add takes for this object: key, or key + "_safe"
};
// memory_only_ok: a module that cannot be written to any cache directory is still returned (it is about to be loaded);
// false = the caller wants the file (exa_compile).
// safe: compile with safe_flags() on top of the base flags (another cache file, another preloaded name: <key>_safe) — what a
// module is rebuilt with when one of its kernels has outgrown the 256 architectural VGPRs (exa_runtime.cpp, audited_code_object)
CodeObject get_code_object(const std::string &source, bool memory_only_ok, bool safe = false);
// the conservative allocator flags ($EXAHIP_SAFE_FLAGS; "none" = never recompile), as one string
std::string safe_flags();
bool cache_add(const std::string &name, const void *blob, size_t len);
bool cache_has(const std::string &name);      // handed over in memory (exa_cache_add)?
std::string writable_cache_dir();     // "" when there is none
// facts about a module that travel with it (exa_build.cpp "notes"): "" when there is none
std::string note_lookup(const std::string &key);
void note_store(const std::string &key, const std::string &note, bool persist);
// every kernel of a code object with its resources, from the AMDGPU metadata note (msgpack) of the ELF (or of the gfx950
// entry of a clang offload bundle); false = the metadata could not be read — callers treat that as "unknown, assume the worst"
struct KernelInfo {
    std::string name;
    int vgpr = 0, agpr = 0, sgpr = 0, scratch = 0, vgpr_spill = 0, sgpr_spill = 0, lds = 0;
    // the kernel lives within the 256 architectural VGPRs: no AGPRs (spill space in a kernel without MFMA), no scratch
    bool fits() const { return agpr == 0 && scratch == 0 && vgpr_spill == 0; }
};
bool code_object_kernels(const std::vector<char> &image, std::vector<KernelInfo> &out);
// registers, scratch bytes per lane, spilled VGPRs and spilled SGPRs of a kernel, from the code object's metadata; false = not found
bool kernel_resources(const std::vector<char> &image, const std::string &kernel, int *vgpr, int *agpr, int *scratch, int *vgpr_spill, int *sgpr_spill = nullptr);

bool tune_lookup(const std::string &key, const std::string &signature, int *value);
void tune_store(const std::string &key, const std::string &signature, int value);

}  // namespace exa
