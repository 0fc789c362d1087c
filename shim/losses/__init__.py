"""Reference package path `losses` (src/losses/)."""
