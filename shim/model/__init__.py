"""Reference package path `model` (src/model/): every module here re-exports its gaviko_amd.model namesake."""
