# bench.jl — SURVEY §8d: the reference's OWN CPU `mul!` timed beside the MI355X extension, in one Julia process, on the
# shared splitmix64 generator, printing the JSON fields of /root/repo/bench.py. NOT EXECUTED in the build image (no
# Julia); every device call below goes through julia/LinearOperatorsMXLOExt.jl, whose ccalls are checked against the
# headers by tests/test_julia_binding.py. On a Julia-equipped MI355X host:
#
#   LD_LIBRARY_PATH=<repo>/linearoperators.jl_amd/csrc julia -t auto --project=<env with LinearOperators> bench.jl [n] [steps] [warmup]
#
# Workload = BASELINE.json configs[1]: opHouseholder(h) 5-arg mul!(res, H, v, 1, 0), n = 10^8 Float64, 40 B/elt
# (16 B/elt dot pass + 24 B/elt update pass). `cpu_baseline.kind` is "reference": the CPU leg IS LinearOperators.jl
# (`mulHouseholder!`, src/linalg.jl:77-83: LinearAlgebra.dot -> OpenBLAS ddot, then one broadcast pass), on
# `BLAS.get_num_threads()` BLAS threads and one Julia thread for the broadcast (Julia broadcast does not thread).
using LinearAlgebra, LinearOperators, Printf
include(joinpath(@__DIR__, "LinearOperatorsMXLOExt.jl"))
using .LinearOperatorsMXLOExt: MXVector, synchronize
const MX = LinearOperatorsMXLOExt

# counter-based generator shared with oracle/lo_oracle_mt.c (orc_mt_fill_f64) and tools/: x[i] = lo + (hi-lo)*u01(seed+i)
@inline function splitmix(x::UInt64)
  x += 0x9E3779B97F4A7C15
  x = (x ⊻ (x >> 30)) * 0xBF58476D1CE4E5B9
  x = (x ⊻ (x >> 27)) * 0x94D049BB133111EB
  x ⊻ (x >> 31)
end
function fill_u01!(x::Vector{Float64}, seed::UInt64, lo::Float64, hi::Float64)
  Threads.@threads for i in eachindex(x)
    @inbounds x[i] = lo + (hi - lo) * (Float64(splitmix(seed + UInt64(i - 1)) >> 11) * (1.0 / 9007199254740992.0))
  end
  x
end

function timed(f, steps, warmup; sync = () -> nothing)
  for _ = 1:warmup
    f()
  end
  sync()
  t0 = time_ns()
  for _ = 1:steps
    f()
  end
  sync()
  (time_ns() - t0) / 1e9 / steps
end

function main()
  n = length(ARGS) >= 1 ? parse(Int, ARGS[1]) : 100_000_000
  steps = length(ARGS) >= 2 ? parse(Int, ARGS[2]) : 200
  warmup = length(ARGS) >= 3 ? parse(Int, ARGS[3]) : 20
  ncpu = min(n, 50_000_000)                      # bounded CPU sample (about 10-30 s of CPU work), stated in the line
  h = fill_u01!(Vector{Float64}(undef, n), 0x000000005EED0001, -0.5, 0.5)
  h ./= norm(h)
  v = fill_u01!(Vector{Float64}(undef, n), 0x000000005EED0002, -1.0, 1.0)

  # ---- device leg: the extension's opHouseholder on device vectors (one ccall per mul!)
  hd, vd = MXVector(h), MXVector(v)
  resd = MXVector{Float64}(undef, n)
  Hd = opHouseholder(hd)
  t_spin = time()
  while time() - t_spin < 0.5                    # clocks up (untimed), as bench.py does
    for _ = 1:20
      mul!(resd, Hd, vd, 1.0, 0.0)
    end
    synchronize()
  end
  sec = timed(() -> mul!(resd, Hd, vd, 1.0, 0.0), steps, warmup; sync = synchronize)
  gbs = 40.0 * n / sec / 1e9

  # ---- parity of the two legs on the SAME data (reduction: 1e-12, DESIGN.md §2)
  Hh = opHouseholder(h)
  resh = similar(v)
  mul!(resh, Hh, v, 1.0, 0.0)
  err = norm(Array(resd) - resh) / norm(resh)

  # ---- CPU leg: the reference itself
  hc, vc, rc = h[1:ncpu], v[1:ncpu], Vector{Float64}(undef, ncpu)
  Hc = opHouseholder(hc)
  mul!(rc, Hc, vc, 1.0, 0.0)                    # warm-up / page-in
  reps = 0
  t0 = time()
  while true
    mul!(rc, Hc, vc, 1.0, 0.0)
    reps += 1
    (time() - t0 > 8.0 || reps >= 20) && break
  end
  csec = (time() - t0) / reps
  cpu = 40.0 * ncpu / csec / 1e9

  @printf("{\"metric\": \"mul! GB/s (frac HBM peak) at n=10^8 fp64; L-BFGS apply/s, 1/2/4/8 GPU\", \"value\": %.1f, \"unit\": \"GB/s\", ", gbs)
  @printf("\"n_gpus\": 1, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"higher_is_better\": true, \"scaling\": \"weak\", ", steps, warmup, sec * 1e3)
  @printf("\"vs_baseline\": null, \"dtype\": \"f64\", \"data\": \"synthetic\", ")
  @printf("\"config\": {\"workload\": \"opHouseholder(h) 5-arg mul!(res,H,v,1,0), n=%d fp64 (configs[1]), Julia host over ccall\", \"n_per_gpu\": %d, \"algorithmic_bytes_per_elt\": 40}, ", n, n)
  @printf("\"frac_hbm_peak\": %.4f, \"parity_rel_l2_vs_reference_cpu\": %.3e, ", gbs / 8000.0, err)
  @printf("\"cpu_baseline\": {\"value\": %.2f, \"unit\": \"GB/s\", \"cores\": %d, \"kind\": \"reference\", ", cpu, BLAS.get_num_threads())
  @printf("\"sample\": \"LinearOperators.jl mulHouseholder! (src/linalg.jl:77-83) on n=%d fp64, %d reps, %.1f ms/apply; ddot on %d BLAS threads, broadcast on 1 Julia thread; host has %d logical CPUs\"}}\n",
          ncpu, reps, csec * 1e3, BLAS.get_num_threads(), Sys.CPU_THREADS)
end

main()
