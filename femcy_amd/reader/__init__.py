from .inp_info import InpInfo
from .inp_info_base import InpInfoBase

__all__ = ["InpInfo", "InpInfoBase"]
