"""Drop-in shim: `from model import VPTREnc, VPTRDec, VPTRFormerNAR, ...` (as train_NAR.py:13-14 / train_FAR.py:13-14
do) resolves to the MI355X-native implementation in vptr_amd.model."""
from vptr_amd.model import *  # noqa: F401,F403
from vptr_amd.model import __all__  # noqa: F401
