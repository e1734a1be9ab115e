"""Import alias: the package directory is ``toolbox-for-asr-and-tts_amd`` (not a valid Python
identifier), so ``import toolbox_for_asr_and_tts_amd`` resolves to it through this shim."""
import os as _os

_real = _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), _os.pardir,
                                        "toolbox-for-asr-and-tts_amd"))
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py"), "r", encoding="utf-8") as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f, _os
