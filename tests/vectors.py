"""Test-side alias of forge_ec_amd.synth (seeded synthetic inputs)."""
from forge_ec_amd.synth import *  # noqa: F401,F403
from forge_ec_amd.synth import _lt, _sub_const  # noqa: F401
