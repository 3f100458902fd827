"""gingr_amd -- MI355X (gfx950) implementation of GiNGR's per-iteration GP-regression update.

Only what the hot path needs: `csrc/` (HIP kernels + the C ABI of include/gingr_hip.h), `_native` (ctypes binding) and
`api` (host-side mirror of the reference's GingrConfig / GingrRegistrationState / GingrAlgorithm plugin surface).
There is no CPU implementation in this package.
"""
from .api import (  # noqa: F401
    Context, CorrespondencePairs, CpdConfiguration, CpdRegistration, CpdRegistrationState, DeviceModel, EulerAngles,
    FittingStatuses, GeneralRegistrationState, GingrAlgorithm, GlobalTranformationType, IcpConfiguration,
    IcpRegistration, IcpRegistrationState, LandmarkCorrespondences, ModelFittingParameters, PointDistributionModel,
    DevicePointDistributionModel, GaussianKernelParameters, GPMMTriangleMesh3D, PointSetHelper, automaticGPMMfromTemplate,
)
from ._native import GingrNativeError  # noqa: F401
from .group import DeviceGroup  # noqa: F401  (in-library multi-GPU group: gingr_group_*)
from . import io  # noqa: F401  (file / wire formats shared with the Scala host)
from . import classic  # noqa: F401  (the reference's other/ CPD family on the device)
from . import sampling  # noqa: F401  (Metropolis-Hastings chain, evaluators, proposals, accuracy metrics)
from . import helper  # noqa: F401  (log -> shapes, posterior variance maps, progress call-back)
from . import simple  # noqa: F401  (GingrInterface / SimpleRegistrator: the entry points of the reference's demos)
from .simple import GingrInterface, SimpleRegistrator, TranslationAfterRotation  # noqa: F401
from .sampling import (  # noqa: F401
    IndependentPointDistanceEvaluator, IndependentPoints, ModelEvaluator, ProbabilisticSettings, RegistrationComparison,
    TriangleMesh3D,
)
