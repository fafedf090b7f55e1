"""Drop-in replacement for the reference's ``models`` package (same module and class names)."""
