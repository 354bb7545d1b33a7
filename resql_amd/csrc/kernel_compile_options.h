// the options every pipeline kernel is compiled with - by the engine in process (runtime.cpp) and by the helper processes
// (kernel_compiler.cpp); they are part of the code-object cache key
#pragma once
#define RSQ_HIPRTC_ARCH "--offload-arch=gfx950"
#define RSQ_HIPRTC_OPT "-O3"
#define RSQ_HIPRTC_STD "-std=c++17"
