"""Modules named after the reference scripts, each exporting that script's ``train`` / ``validate`` (same signatures,
same return tuples) backed by the engine -- `import ssl_cr_histo_amd.scripts.eval_BreastPathQ_SSL_CR as m; m.train(...)`
is the one-line swap for the reference's module-level functions."""
