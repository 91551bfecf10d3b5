"""TEST / BENCH HARNESS (not the engine, never imported by mira_amd/): Python restatements of out-of-scope host code of the
reference that produce INPUTS for the engine's device paths -- the symbolic pipeline behind the cross-term graphs
(expression, grouped_poly, main_gate, graph_evaluator: src/polynomial/*.rs, src/main_gate.rs, src/plonk/util.rs) and
ProtoGalaxy's host arithmetic around the device tree reduction and transforms (protogalaxy: src/nifs/protogalaxy/poly).
The reference's own Display known-answer tests are asserted on these modules (tests/test_cross_term_expressions.py)."""
