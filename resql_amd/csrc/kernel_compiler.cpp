// kernel_compiler.cpp — the helper process that turns generated pipeline sources into gfx950 code objects.
//
// The reference compiles a query in 0.6-3 ms (asmjit, reference src/JitContextFlounder.h:410-456); hiprtc takes 0.2-1.8 s per kernel
// and serialises inside one process (one lock around the compiler).  A statement with eleven kernels (TPC-H Q5) therefore waited
// for the SUM of its compiles.  The engine starts one of these helpers per kernel instead (runtime.cpp compileManyToCache): fresh
// processes that load hiprtc, never touch a GPU, and leave their result in the code-object cache.
//
//   rsq_kernel_compiler <include dir> <cache dir> <arch option> <optimisation option> <language option> <key> [<key> ...]
//
// For every key: reads <cache dir>/<key>.hip, compiles it with the three options the LIBRARY passes (they are part of the cache key the
// library computed: a helper left over from an older build cannot publish code under a key that claims other options), publishes
// <cache dir>/<key>.hsaco by rename.  A failed compile leaves <cache dir>/<key>.err with hiprtc's log; the exit code is the number of failed keys.
#include <hip/hiprtc.h>
#include <unistd.h>

#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>

static bool readFile(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return false;
    std::ostringstream ss; ss << f.rdbuf();
    out = ss.str();
    return true;
}

static bool publish(const std::string& path, const std::string& bytes) {
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    std::ofstream f(tmp, std::ios::binary);
    if (!f.is_open()) return false;
    f.write(bytes.data(), (std::streamsize)bytes.size());
    f.close();
    if (!f || rename(tmp.c_str(), path.c_str()) != 0) { (void)remove(tmp.c_str()); return false; }
    return true;
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: rsq_kernel_compiler <include dir> <cache dir> <arch option> <opt option> <std option> <key>...\n"); return 64; }
    const std::string inc = std::string("-I") + argv[1], cache = argv[2];
    int failed = 0;
    for (int i = 6; i < argc; i++) {
        const std::string base = cache + "/" + argv[i];
        std::string source;
        if (!readFile(base + ".hip", source)) { failed++; continue; }
        hiprtcProgram prog;
        if (hiprtcCreateProgram(&prog, source.c_str(), "rsq_pipeline.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { failed++; continue; }
        const char* opts[] = {argv[3], argv[4], argv[5], inc.c_str()};
        if (hiprtcCompileProgram(prog, 4, opts) != HIPRTC_SUCCESS) {
            size_t n = 0; hiprtcGetProgramLogSize(prog, &n);
            std::string log(n, '\0');
            if (n) hiprtcGetProgramLog(prog, &log[0]);
            (void)publish(base + ".err", log);
            hiprtcDestroyProgram(&prog);
            failed++;
            continue;
        }
        size_t n = 0; hiprtcGetCodeSize(prog, &n);
        std::string code(n, '\0');
        hiprtcGetCode(prog, &code[0]);
        hiprtcDestroyProgram(&prog);
        if (!publish(base + ".hsaco", code)) failed++;
    }
    return failed > 63 ? 63 : failed;
}
