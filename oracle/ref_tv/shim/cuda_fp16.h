// intentionally empty: __half and its conversions come from ../cuda_host_exec.h (force-included)
