"""fgt_amd — MI355X (gfx950) native hot path of hitachinsk/FGT behind the reference's nn.Module API.

    from fgt_amd.fgt_model import Model          # drop-in for FGT.models.model.Model
    fgt_amd/dropin/{FGT,LAFC,RAFT}               # packages with the reference's import paths

All device computation lives in fgt_amd/lib/libfgt_hip.so (sources: fgt_amd/csrc, C ABI: include/fgt_hip.h).
There is no CPU or PyTorch fallback: ops raise if the library is missing or tensors are not on the GPU.
"""
__version__ = "0.1.0"
