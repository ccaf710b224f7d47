"""tokenizer_amd -- MI355X-native Llama-3 encode path (HIP kernels behind the C ABI of include/l3hip.h).

`capi` is the ctypes binding of the C ABI; `llama3` mirrors the reference's Go API
(llama3.New / Encode / Decode / scanner) on top of it.  There is no CPU fallback.
"""
__all__ = ["capi"]
