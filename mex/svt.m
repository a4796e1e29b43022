function X = svt(Y, tau)
% Drop-in for benchmark_algorithms/svt.m.
  X = jstsp_mex('svt', Y, tau);
end
