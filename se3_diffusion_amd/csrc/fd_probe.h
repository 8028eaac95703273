// Gate for the timing / ablation hooks of the product kernel sources (EM_PHASE_TIMING, EM_ABLATE_PQ, EM_PLAIN_SAVES in fd_edge_mlp.hip; the FL_ABL_*
// hooks of fd_ipa_flash.hip).  Several of them produce WRONG RESULTS BY DESIGN (a fetch replaced by a constant, a phase skipped) so
// that its cost can be read off a timing: they exist for tools/probes/* only.  A source that sees one of those macros includes this
// header, and this header refuses to compile unless the build says explicitly that it is a probe build -- a stray or mistyped -D in a
// product build is a compile error, not a silently wrong library.  se3_diffusion_amd/build.py never defines FD_PROBE_BUILD;
// fd_build_flags() (fd_api.hip) reports it, and tests/test_abi.py asserts that the shipped library was built without it.
#pragma once
#ifndef FD_PROBE_BUILD
#error "probe / ablation macro defined in a product build (timing hooks need -DFD_PROBE_BUILD: tools/probes/*)"
#endif
