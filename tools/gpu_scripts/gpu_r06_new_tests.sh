#!/bin/bash
# Round 6: the tests that are new or changed this round, then the whole GPU suite.  Logs land in gpurun_out/r06a/.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06a; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_deviations.py "tests/test_gpu_boundary.py::test_stream_reset_on_every_kind_of_handle" \
    tests/test_gpu_boundary.py::test_set_trace_on_a_handle_delivers_the_character_layers_text \
    tests/test_gpu_boundary.py::test_singleton_prints_the_reference_trace_when_asked \
    tests/test_gpu_boundary.py::test_messages_reach_add_message_without_a_flush \
    tests/test_gpu_boundary.py::test_live_capture_overrun_is_counted_not_silent \
    tests/test_gpu_parity.py::test_noise_only_has_no_near_tie -q -m gpu > $O/new.log 2>&1; echo "new rc=$?"
tail -30 $O/new.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1; rc=$?
tail -8 $O/suite.log; exit $rc
