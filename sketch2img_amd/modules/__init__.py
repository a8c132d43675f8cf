"""Mirror of the reference's ``modules`` package for the hot path (same names, arguments and error
behaviour; compute in libskg.so).  Importable as ``modules.*`` and ``sketch2img.modules.*`` through the alias
packages at the repository root (the reference itself uses both spellings: SURVEY Q12)."""
