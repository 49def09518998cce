// canary_host.cpp — replays ONE recorded kernel launch (exa_debug_dump_window_launch, "EXADUMP1") against a code object, without
// libexahip: the standalone reproducer of the wrong sums an over-sized window kernel returns under the default register
// allocator (profiles/NOTES.md, rounds 3-4).  TEST INFRASTRUCTURE — nothing in the product uses it.
//   hipcc -O2 -o canary_host canary_host.cpp
//   hipcc --genco --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -w [-mllvm -grow-region-complexity-budget=0] -o k.hsaco hprodw_canary.hip
//   ./canary_host hprodw_canary.dump k.hsaco [poison]      exit 0 = equal to the recorded output, 1 = different, 2 = could not run
// "poison": a kernel occupying every VGPR / AGPR of the chip with NaN runs first, so that a kernel reading a register lane it
// never wrote fails every time instead of when stale contents happen to matter.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 2; } } while (0)

// 512 registers per lane (256 VGPRs + 256 AGPRs), all written with a NaN bit pattern; 4096 workgroups cover the chip several times
__global__ void __launch_bounds__(256) poison_registers(double *sink) {
    asm volatile(
        "v_mov_b32 v255, 0x7ff80000\n"
        ".irp r,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127,128,129,130,131,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,150,151,152,153,154,155,156,157,158,159,160,161,162,163,164,165,166,167,168,169,170,171,172,173,174,175,176,177,178,179,180,181,182,183,184,185,186,187,188,189,190,191,192,193,194,195,196,197,198,199,200,201,202,203,204,205,206,207,208,209,210,211,212,213,214,215,216,217,218,219,220,221,222,223,224,225,226,227,228,229,230,231,232,233,234,235,236,237,238,239,240,241,242,243,244,245,246,247,248,249,250,251,252,253,254\n"
        "v_mov_b32 v\\r, v255\n"
        ".endr\n"
        ".irp r,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127,128,129,130,131,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,150,151,152,153,154,155,156,157,158,159,160,161,162,163,164,165,166,167,168,169,170,171,172,173,174,175,176,177,178,179,180,181,182,183,184,185,186,187,188,189,190,191,192,193,194,195,196,197,198,199,200,201,202,203,204,205,206,207,208,209,210,211,212,213,214,215,216,217,218,219,220,221,222,223,224,225,226,227,228,229,230,231,232,233,234,235,236,237,238,239,240,241,242,243,244,245,246,247,248,249,250,251,252,253,254,255\n"
        "v_accvgpr_write_b32 a\\r, v255\n"
        ".endr\n"
        ::: "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
    if (sink && threadIdx.x == 9999) sink[0] = 1.0;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: canary_host DUMP CODE_OBJECT [poison]\n"); return 2; }
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
        else { out_arg = a; out_bytes = n; std::vector<double> nan((size_t)(n / 8), NAN); CHK(hipMemcpy(dev[a], nan.data(), (size_t)n, hipMemcpyHostToDevice)); }
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
    int bad_runs = 0;
    double worst = 0.0;
    long long nbad = 0;
    for (int run = 0; run < 3; run++) {
        if (argc > 3 && !strcmp(argv[3], "poison")) { poison_registers<<<4096, 256>>>(nullptr); CHK(hipDeviceSynchronize()); }
        std::vector<double> nan((size_t)nexp, NAN);
        CHK(hipMemcpy(dev[out_arg], nan.data(), (size_t)out_bytes, hipMemcpyHostToDevice));
        CHK(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, (unsigned)block, 1, 1, (unsigned)lds, nullptr, args.data(), nullptr));
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(got.data(), dev[out_arg], (size_t)out_bytes, hipMemcpyDeviceToHost));
        long long bad = 0;
        for (long long i = 0; i < nexp; i++) {
            if (std::isnan(expect[i]) && std::isnan(got[i])) continue;
            const double d = std::fabs(got[i] - expect[i]) / std::fmax(1.0, std::fabs(expect[i]));
            if (!(d <= 1e-9)) { bad++; if (std::isfinite(d)) worst = std::fmax(worst, d); }
        }
        if (bad) { bad_runs++; nbad = bad; }
    }
    printf("%s: %lld workgroups x %lld lanes, %lld entries: %s (%d of 3 runs differ; last: %lld entries, worst finite relative difference %.3g)\n", kname.c_str(), grid, block,
           nexp, bad_runs ? "DIFFERENT from the recorded output" : "equal to the recorded output", bad_runs, nbad, worst);
    return bad_runs ? 1 : 0;
}
