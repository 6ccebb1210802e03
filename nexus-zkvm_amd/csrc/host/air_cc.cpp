// nx_air_cc: compiles ONE part of a generated AIR / logup source for gfx950 with hiprtc and writes the code object.
// Started by libnexus_hip.so (csrc/air_jit.hip compile_parts) side by side with its siblings: hiprtc serialises the threads of a process,
// not processes.  No GPU is touched.
// usage: nx_air_cc <source file> <output code object> <optimisation level, e.g. -O1> [<path of the hiprtc library to use>]
// The library passes the path of the hiprtc IT is linked against (a process may carry another one than /opt/rocm's — PyTorch ships its
// own), so a part compiled here is byte for byte what the library's own hiprtc_compile would have produced.
// exit 0: the output file holds the code object; otherwise <output>.log holds the compiler's log and the library compiles the part itself
// (so the error text reaches the caller the usual way).
#include <dlfcn.h>
#include <stdio.h>
#include <string>
#include <vector>

typedef void* Program;
typedef int (*CreateFn)(Program*, const char*, const char*, int, const char* const*, const char* const*);
typedef int (*CompileFn)(Program, int, const char* const*);
typedef int (*SizeFn)(Program, size_t*);
typedef int (*DataFn)(Program, char*);
typedef int (*DestroyFn)(Program*);

int main(int argc, char** argv) {
    if (argc != 4 && argc != 5) { fprintf(stderr, "usage: nx_air_cc <source> <output> <-O level> [<libhiprtc path>]\n"); return 2; }
    void* lib = dlopen(argc == 5 && argv[4][0] ? argv[4] : "libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "nx_air_cc: %s\n", dlerror()); return 8; }
    CreateFn create = (CreateFn)dlsym(lib, "hiprtcCreateProgram");
    CompileFn compile = (CompileFn)dlsym(lib, "hiprtcCompileProgram");
    SizeFn log_size = (SizeFn)dlsym(lib, "hiprtcGetProgramLogSize"), code_size = (SizeFn)dlsym(lib, "hiprtcGetCodeSize");
    DataFn get_log = (DataFn)dlsym(lib, "hiprtcGetProgramLog"), get_code = (DataFn)dlsym(lib, "hiprtcGetCode");
    DestroyFn destroy = (DestroyFn)dlsym(lib, "hiprtcDestroyProgram");
    if (!create || !compile || !log_size || !code_size || !get_log || !get_code || !destroy) return 9;
    std::string src;
    {
        FILE* f = fopen(argv[1], "rb");
        if (!f) return 3;
        char buf[1 << 16]; size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) src.append(buf, got);
        fclose(f);
    }
    Program rp = nullptr;
    if (create(&rp, src.c_str(), "nx_air_kernel.hip", 0, nullptr, nullptr) != 0) return 4;
    const char* opts[] = {"--offload-arch=gfx950", argv[3]};
    if (compile(rp, 2, opts) != 0) {
        size_t ls = 0; (void)log_size(rp, &ls);
        std::string log(ls, '\0'); if (ls) (void)get_log(rp, &log[0]);
        FILE* f = fopen((std::string(argv[2]) + ".log").c_str(), "wb");
        if (f) { fwrite(log.data(), 1, log.size(), f); fclose(f); }
        return 5;
    }
    size_t cs = 0; (void)code_size(rp, &cs);
    std::vector<char> code(cs);
    (void)get_code(rp, code.data());
    (void)destroy(&rp);
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 6;
    const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
    return fclose(f) == 0 && ok ? 0 : 7;
}
