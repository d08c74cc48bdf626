"""MI355X-native Eta-Inversion engine: ctypes binding (`_capi`), engine wrapper (`engine`), weights (`weights`).
The compute path is libetainv_hip.so (hand-written HIP for gfx950); there is no CPU fallback."""
