function [IDX, C, SUMD, D, OUTPUT, C2, IDX2, D2, SUMD2] = kmeans_sparsified(X, K, varargin)
%KMEANS_SPARSIFIED  K-means on preconditioned + sparsified data, Lloyd iterations on an AMD MI355X.
%
%   [IDX, C, SUMD, D, OUTPUT] = kmeans_sparsified(X, K, 'Name', value, ...)
%   [..., C_twoPass, IDX_twoPass, D_twoPass, SUMD_twoPass] = kmeans_sparsified(...)
%
% MATLAB host for libspkm.so (include/spkm.h).  Same entry point, option names, defaults, outputs and error
% conditions as kmeans_sparsified.m of stephenbeckr/SparsifiedKMeans v2.1 (cited below as REF:line), written from
% scratch around the GPU engine.  With 'Sparsify',true and the Hadamard sketch the whole pipeline runs on the device:
%   * spkm_lloyd('sparsify', X, p2, d, s, seed): X*(1+2eps), zero-pad, D, FWHT / sqrt(p2) and the s sampled rows per
%     column in ONE fused pass (REF:292-334) -- the dense data crosses PCIe once, the sparse p2 x n matrix never exists on
%     the host;
%   * seeding: spkm_lloyd('kpp', K, gam) (k-means++ with a running minimum on the device, MATLAB's rand as the random
%     source) or spkm_lloyd('columns', randsample(n, K)) ('sample');
%   * every iteration is ONE mex call that returns centers, dff, obj and the cluster sizes -- nothing of size n;
%   * IDX and D are fetched ONCE per replicate after the loop (spkm_lloyd('assignments'), spkm_lloyd('distances', ...)),
%     together with the objective of the last iteration (REF:471 evaluates obj every iteration but uses it only for
%     Display = 'iter' and after the loop: unless it is displayed per iteration the library may leave it out).
% Other sketches ('DCT', function handles), 'DataFile' and 'MLcorrection',false keep the reference's host-side route and
% upload the sparse result (spkm_lloyd('upload', X)); iterations with sparse centres go through findClusterAssignments,
% whose mex (SparseMatrixMinusCluster) is this repository's GPU gateway as well.
%
% STATUS: NOT RUN IN THIS REPOSITORY -- neither MATLAB nor Octave exists in its build / test environment, so this
% file and the gateways in matlab/*.c are written against the documented mex / MATLAB API and have never been
% executed.  The same control flow is what sparsifiedkmeans_amd/kmeans.py implements and what the GPU test-suite
% exercises through the identical C ABI.  Needs on the MATLAB path: the reference's private/ helpers
% (findClusterAssignments, Arthur_initialization, randsample_fixedNumberEntries, sampleAndMixFromLargeFile,
% recalculateAssignmentLargeFile) and the gateways of matlab/ built as in INTEGRATION.md section 2.
%
% Options (REF:130-155), defaults in brackets:
%   'Replicates' [1]  'Start' ['Arthur' | 'sample' | 'uniform' | '++' | 'kmeans++' | K x p matrix]  'MaxIter' [100]
%   'Display' ['off' | 'iter' | 'final']  'PrintEvery' [10]  'Tol' [1e-6]  'Sparsify' [false]  'SparsityLevel' [0.01]
%   'SketchType' ['auto' | 'Hadamard' | 'DCT' | 'none' | {H, Ht}]  'EmptyAction' ['singleton' | 'error' | 'drop']
%   'ColumnSamples' [false]  'MLcorrection' [true]  'DataFile' []  'MB_limit' [500]  'DataFileVerbose' [false]
%   'SparsityIgnoreUpsampling' [false]  'FORCE_BUG' [false]  'tryBuiltinMex' [true]  'unbiasedDistance' [true]
%   'unbiasedInitialization' [true]  'denseCenters' [false]

tStart = tic;
ip = inputParser;
ip.addParameter('Replicates', 1);
ip.addParameter('Start', 'Arthur');
ip.addParameter('MaxIter', 100);
ip.addParameter('Display', 'off', @(s) any(strcmpi(s, {'off', 'iter', 'final'})));
ip.addParameter('PrintEvery', 10);
ip.addParameter('Tol', 1e-6);
ip.addParameter('Sparsify', false);
ip.addParameter('SparsityLevel', 0.01, @(g) g > 0 && g <= 1);
ip.addParameter('SketchType', 'auto');
ip.addParameter('EmptyAction', 'singleton', @(s) any(strcmpi(s, {'singleton', 'error', 'drop'})));
ip.addParameter('ColumnSamples', false);
ip.addParameter('MLcorrection', true);
ip.addParameter('DataFile', []);
ip.addParameter('MB_limit', 500);
ip.addParameter('DataFileVerbose', false);
ip.addParameter('SparsityIgnoreUpsampling', false);
ip.addParameter('FORCE_BUG', false);
ip.addParameter('tryBuiltinMex', true);
ip.addParameter('unbiasedDistance', true);
ip.addParameter('unbiasedInitialization', true);
ip.addParameter('denseCenters', false);
ip.parse(varargin{:});
o = ip.Results;
o.MLcorrection = o.MLcorrection && o.Sparsify;                                  % REF:171
if ischar(X), o.DataFile = X; X = []; end                                       % REF:179-183
fromDisk = ~isempty(o.DataFile);
OUTPUT = struct('LoadFromDisk', fromDisk, 'Options', o);
show = @(lvl) any(strcmpi(o.Display, lvl));
haveEngine = exist('spkm_lloyd', 'file') == 3;

% ---- data, orientation (points become COLUMNS), sizes ----
if fromDisk
    if ~exist(o.DataFile, 'file') && ~exist([o.DataFile '.mat'], 'file')
        error('kmeans_sparsified:noFile', 'Cannot find specified data file to load');
    end
    t1 = tic; [pp, nn] = sampleAndMixFromLargeFile(o.DataFile, 0, [], [], 'ColumnSamples', o.ColumnSamples);
    p = pp; n = nn; OUTPUT.TimeToReadSizeOfFile = toc(t1);                      % REF:208-212 (size probe)
else
    if ~o.ColumnSamples, X = X.'; end
    [p, n] = size(X);
end
if n < K, error('kmeans_sparsified:badDimensions', 'X must have more samples than the number of clusters.'); end

% ---- preconditioner and sparsifier (REF:224-358) ----
gam = o.SparsityLevel;  p2 = p;  XFull = [];  onDevice = false;
mixf = @(Z) Z;  unmixf = @(Z) Z;
if o.Sparsify
    sk = o.SketchType;
    if ischar(sk) && strcmpi(sk, 'auto')
        if p == 2^nextpow2(p), sk = 'Hadamard'; else, sk = 'DCT'; end
        OUTPUT.SketchType = sk;
    end
    if show('iter') || show('final'), fprintf('Randomly mixing of type %s\n', char(string(sk))); end
    pad = @(Z) Z;  crop = @(Z) Z;
    if iscell(sk)
        if ~(isa(sk{1}, 'function_handle') && isa(sk{2}, 'function_handle'))
            error('If SketchType is a cell, then both entries should be function handles for forward and adjoint transform');
        end
        H = sk{1};  Ht = sk{2};
    elseif strcmpi(sk, 'Hadamard')
        p2 = 2^nextpow2(p);
        if p < p2, pad = @(Z) [Z; zeros(p2 - p, size(Z, 2))];  crop = @(Z) Z(1:p, :); end
        if exist('hadamard', 'file') ~= 3
            error('kmeans_sparsified:noMex', 'the hadamard mex gateway of this repository is not on the path');
        end
        H = @(Z) hadamard(Z) / sqrt(p2);  Ht = H;  OUTPUT.SlowHadamard = false;   % GPU transform behind the mex name
    elseif strcmpi(sk, 'DCT')
        H = @(Z) dct(Z);  Ht = @(Z) idct(Z);
    elseif any(strcmpi(sk, {'none', 'Nothing'}))
        H = @(Z) Z;  Ht = H;
    else
        error('bad type for "SketchType"');
    end
    if ischar(sk) && any(strcmpi(sk, {'none', 'Nothing'}))
        flip = @(Z) Z;
    else
        if o.FORCE_BUG, d = sign(rand(p2, 1)); else, d = sign(randn(p2, 1)); end  % REF:283-287
        DD = spdiags(d, 0, p2, p2);  flip = @(Z) DD * Z;
    end
    mixf = @(Z) H(flip(pad(Z)));
    unmixf = @(Z) crop(flip(Ht(Z)));
    if fromDisk
        t1 = tic;
        [X, tLoad, tMix, tSample] = sampleAndMixFromLargeFile(o.DataFile, gam, mixf, p2, ...
            'ColumnSamples', o.ColumnSamples, 'MB_limit', o.MB_limit, 'Verbose', o.DataFileVerbose);
        OUTPUT.TimeToReadAndSketchFile = toc(t1);
        OUTPUT.TimeToSketch = tMix;  OUTPUT.TimeToSample = tSample;  OUTPUT.TimeToRead = tLoad;
    else
        if nargout > 5, XFull = X; end
        if ~isreal(X), error('Code and distance computations require real data'); end
        small_p = max(1, round(gam * p2));
        gam = small_p / p;                                                      % REF:329 (divides by p, not p2)
        % (Start = 'uniform' draws between min(X(:)) and max(X(:)) of the SPARSIFIED data (REF:372-374): that needs the sparse
        %  matrix on the host, so such a run keeps the host route -- as the reference runs it)
        uniformStart = ischar(o.Start) && strcmpi(o.Start, 'uniform');
        onDevice = haveEngine && o.MLcorrection && ischar(sk) && strcmpi(sk, 'Hadamard') && ~issparse(X) && p2 <= 65536 && ~uniformStart;
        if onDevice
            % REF:292 (X*(1+2*eps)), :295 (mix), :334 (randsample_fixedNumberEntries) as one fused device pass; the
            % sampled rows come from a counter-based generator keyed by (seed, column) -- any exact without-replacement
            % sampler has the reference's distribution (randsample_block.m:44-84)
            t1 = tic;
            spkm_lloyd('sparsify', full(X), p2, d, small_p, randi(2^31 - 1));
            OUTPUT.TimeToSketch = toc(t1);  OUTPUT.TimeToSample = 0;
            X = [];                                                             % the sparse data lives on the GPU only
        else
            X = X * (1 + 2 * eps);                                              % REF:292
            t1 = tic;  X = mixf(X);  OUTPUT.TimeToSketch = toc(t1);
            t1 = tic;  X = randsample_fixedNumberEntries(X, small_p);  OUTPUT.TimeToSample = toc(t1);
        end
    end
    if show('iter') || show('final')
        if onDevice, fprintf('Randomly taking %.1f%% of the data\n', 100 * gam);
        else, fprintf('Randomly taking %.1f%% of the data; actual dataset is %.1f%% sparse\n', 100 * gam, 100 * nnz(X) / numel(X)); end
    end
    if o.MLcorrection && ~onDevice, Nmask = spones(X); end                      % REF:352-355
elseif fromDisk
    error('kmeans_sparsified:needSparsify', '''DataFile'' is only read on the ''Sparsify'',true path');
end
useEngine = onDevice || (haveEngine && issparse(X) && o.MLcorrection);
if useEngine
    if ~onDevice, spkm_lloyd('upload', X); end
    cleanupObj = onCleanup(@() spkm_lloyd('release')); %#ok<NASGU>
end
lazyObj = ~show('iter');       % obj is displayed per iteration only under Display = 'iter' (REF:472-475)

if ischar(o.Start) && strcmpi(o.Start, 'uniform')
    mn = full(min(X(:)));  mx = full(max(X(:)));
end
if o.Sparsify && o.unbiasedDistance
    findClusters = @(Z, ctr) findClusterAssignments(Z, ctr, o.tryBuiltinMex, gam);
else
    findClusters = @(Z, ctr) findClusterAssignments(Z, ctr, o.tryBuiltinMex);
end

R = o.Replicates;
OUTPUT.iterations = zeros(1, R);  OUTPUT.stoppingDiff = zeros(1, R);  OUTPUT.objectives = zeros(1, R);
OUTPUT.replicateTimes = zeros(1, R);  OUTPUT.replicateTimesJustInitialization = zeros(1, R);
bestObj = Inf;  bestA = [];  bestD = [];  bestC = [];  distances = [];
for trial = 1:R
    t1 = tic;
    % ---- start (REF:381-415) ----
    if ischar(o.Start)
        switch lower(o.Start)
            case 'sample'
                if onDevice, centers = spkm_lloyd('columns', randsample(n, K)); else, centers = X(:, randsample(n, K)); end
            case 'uniform', centers = (mx - mn) * rand(p2, K) - mn;             % (the reference subtracts mn)
            case {'arthur', '++', 'kmeans++', 'k-means++', 'k-means-++'}
                if onDevice
                    if o.unbiasedInitialization, centers = spkm_lloyd('kpp', K, gam); else, centers = spkm_lloyd('kpp', K, []); end
                elseif o.Sparsify && o.unbiasedInitialization
                    centers = Arthur_initialization(X, K, gam);
                else
                    centers = Arthur_initialization(X, K);
                end
            otherwise, error('cannot handle other types of "Start" values');
        end
    else
        S = o.Start;  if ~o.ColumnSamples, S = S.'; end
        centers = mixf(S);
        if R > 1
            warning('kmeans_sparsified:deterministicCenters', 'initialization is specified, so running more than 1 replicate is not helpful');
        end
    end
    if o.denseCenters, centers = full(centers); end
    OUTPUT.replicateTimesJustInitialization(trial) = toc(t1);
    if useEngine, spkm_lloyd('reset'); end                                      % nothing learned carries over

    % ---- Lloyd iterations (REF:417-486) ----
    fetched = false;               % IDX / D of the latest iteration are on the host
    for its = 1:o.MaxIter
        old = centers;  dropList = [];
        if useEngine && (onDevice || ~issparse(centers))
            % one call: assignment + accumulation + centre update + dff + obj on the GPU; four small outputs.  Sparse
            % centres (the iteration after a 'sample' / k-means++ start) take the sparse-centres branch on the device too;
            % the updated centres come back full (ML-corrected columns are > 99 % filled: REF:460-464)
            [centers, ~, obj, nk] = spkm_lloyd('iterate', centers, gam, o.unbiasedDistance, lazyObj);
            emptyK = find(nk == 0);  fetched = false;
        else
            [assignments, distances] = findClusters(X, centers);
            if ~isreal(distances), error('Distance estimates are complex, something went wrong'); end
            if any(distances < 0), error('Found negative distance estimates, something went wrong'); end
            obj = sqrt(sum(distances .^ 2));  fetched = true;
            emptyK = [];
            for k = 1:K
                members = find(assignments == k);
                if isempty(members)
                    emptyK(end + 1) = k; %#ok<AGROW>
                elseif o.MLcorrection
                    centers(:, k) = gam * full(sum(X(:, members), 2)) ./ (full(sum(Nmask(:, members), 2)) + 1e-16);
                else
                    centers(:, k) = mean(full(X(:, members)), 2);
                end
            end
        end
        for k = emptyK(:).'                                                     % REF:432-445
            warning('kmeans_sparsified:dropCluster', 'cluster has lost all its members');
            switch lower(o.EmptyAction)
                case 'singleton'
                    if ~fetched                                                 % this iteration's distances, now (REF:436)
                        [distances, obj] = spkm_lloyd('distances', old, gam, o.unbiasedDistance);
                        assignments = spkm_lloyd('assignments');  fetched = true;
                    end
                    [~, far] = max(distances);
                    if onDevice, centers(:, k) = full(spkm_lloyd('columns', far)); else, centers(:, k) = X(:, far); end
                case 'error',     error('One cluster lost all its members');
                case 'drop',      dropList(end + 1) = k; %#ok<AGROW>
            end
        end
        if ~isempty(dropList)                                                   % REF:454-459
            if ~fetched && useEngine                                            % `distances` / obj of THIS iteration stay (REF:471)
                [distances, obj] = spkm_lloyd('distances', old, gam, o.unbiasedDistance);  fetched = true;
            end
            keep = setdiff(1:K, dropList);
            centers = centers(:, keep);  old = old(:, keep);  assignments = [];  K = numel(keep);
        end
        if issparse(centers) && nnz(centers) / numel(centers) > .99, centers = full(centers); end
        if ~isreal(centers), error('Found complex numbers in centers, something went wrong'); end
        dff = norm(old - centers, 'fro');
        if show('iter') && ~mod(its, o.PrintEvery)
            fprintf('Iter: %3d; change in cluster centers: %.2e; objective: %.2e\n', its, dff, obj);
        end
        if dff < o.Tol, break; end
        if any(isnan(centers(:))), error('Found NaN in centers'); end
    end
    if useEngine && ~fetched
        % IDX, D and the objective of the iteration that turned out to be the last (REF:420,471): fetched once per replicate.
        % `old` holds the centres that iteration's assignment was computed with.
        [distances, obj] = spkm_lloyd('distances', old, gam, o.unbiasedDistance);
        assignments = spkm_lloyd('assignments');
    end
    OUTPUT.replicateTimes(trial) = toc(t1);
    OUTPUT.stoppingDiff(trial) = dff;  OUTPUT.objectives(trial) = obj;  OUTPUT.iterations(trial) = its;
    isBest = obj < bestObj;
    if isBest, bestObj = obj;  bestA = assignments;  bestD = distances;  bestC = centers; end
    if show('iter') || (show('final') && isBest)
        fprintf('Trial %3d of %3d total, objective %.2e\n', trial, R, obj);
    end
end
OUTPUT.TimeInitialization = sum(OUTPUT.replicateTimesJustInitialization);
OUTPUT.TimeAlgo_wo_initialization = sum(OUTPUT.replicateTimes) - OUTPUT.TimeInitialization;
SUMD = zeros(K, 1);
for k = 1:K, SUMD(k) = sum(distances(:, bestA == k) .^ 2); end                  % REF:514-518 (last trial's distances)
OUTPUT.TimeOverall_OnePass = toc(tStart);

% ---- outputs in the original coordinates; optional second pass over the unsampled data (REF:522-584) ----
if o.Sparsify, bestC = unmixf(full(bestC)); end
C2 = [];  IDX2 = [];  D2 = [];  SUMD2 = [];
if nargout > 5
    if ~o.Sparsify
        warning('kmeans_sparsified:twoPass', 'There is no sparsification, so the twoPass variables are the same');
        C2 = bestC;  IDX2 = bestA;  D2 = bestD;  SUMD2 = SUMD;
    elseif fromDisk
        warning('kmeans_sparsified:twoPass', 'Requires a second pass over the dataset');
        t1 = tic;
        [IDX2, D2, C2, tRead] = recalculateAssignmentLargeFile(o.DataFile, bestC, bestA, ...
            'ColumnSamples', o.ColumnSamples, 'MB_limit', o.MB_limit);
        OUTPUT.TimeSecondPass_Overall = toc(t1);  OUTPUT.TimeSecondPass_JustRead = tRead;
    else
        t1 = tic;  C2 = zeros(p, K);
        for k = 1:K
            members = find(bestA == k);
            if ~isempty(members), C2(:, k) = mean(full(XFull(:, members)), 2); end
        end
        OUTPUT.TimeSecondPass_Centers = toc(t1);
        if nargout > 6
            t1 = tic;  [IDX2, D2] = findClusters(full(XFull), bestC);  OUTPUT.TimeSecondPass_Assignments = toc(t1);
        end
    end
    if nargout >= 9 && ~isempty(IDX2) && o.Sparsify
        t1 = tic;  SUMD2 = zeros(K, 1);
        for k = 1:K, SUMD2(k) = sum(distances(:, IDX2 == k) .^ 2); end          % REF:564-568 (one-pass distances: kept)
        OUTPUT.TimeSecondPass_SUMD = toc(t1);
    end
end
IDX = bestA(:);  D = bestD(:);
if ~o.ColumnSamples, C = bestC.';  if ~isempty(C2), C2 = C2.'; end, else, C = bestC; end
OUTPUT.TimeOverall = toc(tStart);
end
