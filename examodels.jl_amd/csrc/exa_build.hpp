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
};
// memory_only_ok: a module that cannot be written to any cache directory is still returned (it is about to be loaded);
// false = the caller wants the file (exa_compile).
CodeObject get_code_object(const std::string &source, bool memory_only_ok);
bool cache_add(const std::string &name, const void *blob, size_t len);
std::string writable_cache_dir();     // "" when there is none
// facts about a module that travel with it (exa_build.cpp "notes"): "" when there is none
std::string note_lookup(const std::string &key);
void note_store(const std::string &key, const std::string &note, bool persist);
// registers, scratch bytes per lane, spilled VGPRs and spilled SGPRs of a kernel, from the code object's metadata; false = not found
bool kernel_resources(const std::vector<char> &image, const std::string &kernel, int *vgpr, int *agpr, int *scratch, int *vgpr_spill, int *sgpr_spill = nullptr);

bool tune_lookup(const std::string &key, const std::string &signature, int *value);
void tune_store(const std::string &key, const std::string &signature, int value);

}  // namespace exa
