"""Reference package path `data` (src/data/)."""
