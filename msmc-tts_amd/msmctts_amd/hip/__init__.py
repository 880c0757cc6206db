"""ctypes bindings of libmsmc_hip.so (C ABI: include/msmc_hip.h) and the autograd wrappers over them."""
