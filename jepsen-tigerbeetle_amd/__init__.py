"""MI355X-native linearizability checker behind the knossos / jepsen.checker
surface.  Host side is Python here because this environment has
no JVM; the Clojure/JNA binding is in INTEGRATION.md."""
from . import _native  # noqa: F401
from ._native import NoDeviceError, TbcError, build  # noqa: F401
