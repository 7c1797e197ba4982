"""hirl4ucav_amd — MI355X-native hot path of HIRL4UCAV: batched pursuit-lock-launch env step and the HIRL
(TD3+BC) update as hand-written HIP for gfx950 behind the reference's env / agent API.  See DESIGN.md."""
__version__ = "0.1.0"
