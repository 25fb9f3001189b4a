"""Bench-only build variants of libgmamd.so (python -m generativemodels_amd._build --variant NAME -> lib/libgmamd_NAME.so, loaded through
GM_NATIVE_LIB by the tools that need them): extra defines per variant.  The product library is never built with any of them."""
VARIANTS = {
    "timeline": ["-DGM_CONV_TIMELINE"],                      # tools/conv_timeline.py: cycle stamps per phase of the LDS-DMA convolution
    "ablate": ["-DGM_CONV_ABLATE"],                          # tools/ablate_conv.py: operand traffic / epilogue removable by debug flags
    "timeline_ablate": ["-DGM_CONV_TIMELINE", "-DGM_CONV_ABLATE"],
    "eb": ["-DGM_CONV_EARLY_BARRIER"],                       # round 3 A/B: the tap-group barrier one tap earlier (conv_dma.hip)
    "eb_timeline": ["-DGM_CONV_EARLY_BARRIER", "-DGM_CONV_TIMELINE"],
    # AddressSanitizer build (tools/gpu_asan.sh): device + host code instrumented, kernels for gfx950:xnack+ (run with HSA_XNACK=1,
    # LD_PRELOAD of libclang_rt.asan-x86_64.so, PYTORCH_NO_CUDA_MEMORY_CACHING=1 so that every tensor is its own hipMalloc with red zones).
    # The LDS-DMA requests are inline assembly and are NOT instrumented; every compiler-generated global access is.
    "asan": ["-fsanitize=address", "-shared-libsan", "-g1", "-DGM_ASAN_BUILD"],
    "ldsep": ["-DGM_CONV_LDS_EPILOGUE"],                       # round 5 A/B: the LDS-transposed epilogue on the 64-channel tiles (the pre-round-5 form)
    "ldsep_timeline": ["-DGM_CONV_LDS_EPILOGUE", "-DGM_CONV_TIMELINE"],
    "ebi": ["-DGM_CONV_EARLY_BARRIER", "-DGM_CONV_DMA_INTERLEAVE"],  # ... plus the panel request's DMA instructions spread over a tap's MFMAs
}
