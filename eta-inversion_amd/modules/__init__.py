"""Registries of the reference's modules/__init__.py:31-111: load_inverter / load_editor / get_inversion_methods /
get_edit_methods / register_editor.  Built on the MI355X engine: `etainv`, `dirinv` (+ the plain `diffinv` base) and the
`simple`, `ptp`, `masactrl` editors; the reference's other method names are listed but raise a clear error."""
from functools import partial
from typing import Callable, List

from .inversion.diffusion_inversion import DiffusionInversion
from .inversion.eta_inversion import EtaInversion
from .inversion.direct_inversion import DirectInversion
from .editing.editor import Editor
from .editing.simple_editor import SimpleEditor
from .editing.ptp_editor import PromptToPromptEditor
from .editing.masactrl_editor import MasactrlEditor
from .models import StablePreprocess, StablePostProc, load_diffusion_model


def _not_built(name, *a, **k):
    raise NotImplementedError(f"'{name}' is outside the MI355X hot path built so far (SURVEY.md 8f); available: etainv, dirinv, diffinv / "
                              f"simple, ptp, masactrl")


_inverters = {"diffinv": DiffusionInversion, "etainv": EtaInversion, "dirinv": DirectInversion,
              **{n: partial(_not_built, n) for n in ("nti", "npi", "proxnpi", "edict", "ddpminv", "cyclediff", "regdiffinv")}}
_editors = {"simple": SimpleEditor, "ptp": PromptToPromptEditor, "masactrl": MasactrlEditor,
            **{n: partial(_not_built, n) for n in ("pnp", "pix2pix_zero", "invedit")}}


def register_editor(name: str, editor_cls: Callable) -> None:
    print(f"Registering editor {name}")
    _editors[name] = editor_cls


def get_inversion_methods() -> List[str]:
    return list(_inverters.keys())


def get_edit_methods() -> List[str]:
    return list(_editors.keys())


def load_inverter(type: str, **kwargs) -> DiffusionInversion:
    return _inverters[type](**kwargs)


def load_editor(type: str, **kwargs) -> Editor:
    return _editors[type](**kwargs)
