"""Re-statement (from call sites) of the subset of eladhoffer/utils.pytorch the reference's hot path
imports.  Oracle-side only; see ../README.md."""
