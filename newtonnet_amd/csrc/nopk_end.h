#if defined(__HIP_DEVICE_COMPILE__) && !defined(NNHIP_PACKED_FP32)
#pragma clang attribute pop
#endif
