"""Host-side mirror of the reference's backend interface (include/Config.hpp:11-60,
src/Config.cpp:13-110): same method names, argument meaning and error behaviour, so code and
tests written against `Config` read the same.  The C++ original of this adapter is
adapter/HipConfig.{hpp,cpp}; INTEGRATION.md shows how the reference host links it."""
import enum
import time

import numpy as np

from . import api, host


class SupportType(enum.Enum):       # Config.hpp:44-48
    OpenCL = 0
    Cm = 1
    Default = 2                      # the free slot: framework ID 2 = this HIP backend


class MemType(enum.Enum):           # Config.hpp:51-55
    Buffer = 0
    SVM = 1
    UserProvidedZeroCopy = 2


def selectType(id):                  # Config.cpp:99-110
    if id == 0:
        return SupportType.OpenCL
    if id == 1:
        return SupportType.Cm
    return SupportType.Default


class HipConfig:
    """`Config` implemented on rt_ctx.  updateRendering() = one pass, as Config.cpp:73-91."""

    def __init__(self, width, height, device=0):
        self.mWidth, self.mHeight = width, height
        self.mCurrentSample = 0
        self._ctx = api.RtContext(width, height, device=device)        # allocateBuffer
        self._pixels = np.zeros(width * height, np.uint32)             # pPixels: stable address
        self._cam = np.zeros(api.CAMERA_FLOATS, np.float32)
        self._caption = None

    def sceneSetup(self, spheres, orig, target):                        # OpenCLConfig.cpp:720-747
        self._ctx.set_scene(spheres)
        self._cam[0:3] = orig
        self._cam[3:6] = target

    def updateCamera(self):                                             # OpenCLConfig.cpp:386-392
        self._cam = host.compute_camera(self._cam[0:3], self._cam[3:6], self.mWidth, self.mHeight)
        self._ctx.set_camera(self._cam)

    def getPixels(self):
        return self._pixels

    def setCaptionBuffer(self, buffer):
        self._caption = buffer

    def updateRendering(self, passes=1):                                # Config.cpp:73-91
        start = time.time()
        self._pixels[:] = self._ctx.render_pass(passes)                 # setArguments + execute
        self.mCurrentSample += passes
        elapsed = max(time.time() - start, 1e-9)
        rate = passes * self.mHeight * self.mWidth / elapsed
        text = "Rendering time %.3f sec (pass %d)  Sample/sec  %.1fK\n" % (elapsed, self.mCurrentSample,
                                                                          rate / 1000.0)
        if self._caption is not None:
            self._caption[:] = [text]
        return text

    def close(self):                                                    # freeBuffer
        self._ctx.close()


def createConfig(width, height, frameworkType, useGPU=True, memType=MemType.Buffer):
    """Config.cpp:13-68.  Only the slot the reference leaves unimplemented is served here."""
    if frameworkType is SupportType.Default:
        if memType is not MemType.Buffer:
            raise RuntimeError("Unsupported Memory Type")
        return HipConfig(width, height)
    raise RuntimeError("Unsupported Framework Type")
