// gkjitc -- out-of-process hiprtc for the plan-specialised builds of the dominant kernel (kernels.hip jit_compile).
//
// Why a process: a policy set of several plan groups needs several kernels at its first sweep, and hiprtc compiles of different
// programs only run side by side when the code-object manager underneath allows it -- the one PyTorch-ROCm bundles (and a host
// process that imported torch has loaded) serialises them (measured: four plan groups 17 s "in parallel" against 7.7 s with the
// ROCm 7.2 library alone).  A helper process per build has the compiler to itself: N builds take the time of the slowest.
//   usage: gkjitc <source file> <code object file>        exit 0 = written (write-then-rename); 1 = failed, log on stderr
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: gkjitc <source> <code object>\n"); return 2; }
  std::string src;
  {
    FILE* f = fopen(argv[1], "rb");
    if (!f) { fprintf(stderr, "gkjitc: cannot read %s\n", argv[1]); return 1; }
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) src.append(buf, n);
    fclose(f);
  }
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, src.c_str(), "gk_plan.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { fprintf(stderr, "gkjitc: hiprtcCreateProgram failed\n"); return 1; }
  const char* opt_level = getenv("GK_JIT_OPT") ? getenv("GK_JIT_OPT") : "-O3";   // (diagnostic aid, e.g. -O1: is a device-only difference the optimiser's? use with GK_JIT_CACHE_DIR=off -- the cache is keyed by the text)
      const char* opts[] = {"--offload-arch=gfx950", opt_level, "-std=c++17"};   // (the options of the in-process build, kernels.hip)
  if (hiprtcCompileProgram(prog, 3, opts) != HIPRTC_SUCCESS) {
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls, 0);
    if (ls) hiprtcGetProgramLog(prog, &log[0]);
    fprintf(stderr, "%s\n", log.substr(0, 4000).c_str());
    return 1;
  }
  size_t cs = 0;
  if (hiprtcGetCodeSize(prog, &cs) != HIPRTC_SUCCESS || cs == 0) return 1;
  std::vector<char> code(cs);
  if (hiprtcGetCode(prog, code.data()) != HIPRTC_SUCCESS) return 1;
  const std::string tmp = std::string(argv[2]) + ".part";
  FILE* o = fopen(tmp.c_str(), "wb");
  if (!o) { fprintf(stderr, "gkjitc: cannot write %s\n", tmp.c_str()); return 1; }
  const bool ok = fwrite(code.data(), 1, cs, o) == cs;
  fclose(o);
  if (!ok || rename(tmp.c_str(), argv[2]) != 0) { remove(tmp.c_str()); return 1; }
  return 0;
}
