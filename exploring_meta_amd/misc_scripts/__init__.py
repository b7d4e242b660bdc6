"""Counterparts of the reference's misc_scripts that drive the learner step by step (continual-learning accuracy matrix,
representation change); the analysis / plotting around them (CCA, CKA, plots, result files) stays with the caller."""
