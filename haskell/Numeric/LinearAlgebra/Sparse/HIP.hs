{-# LANGUAGE ForeignFunctionInterface, FlexibleContexts #-}
-- | MI355X backend for the SpMV / CGS / BiCGSTAB / Arnoldi hot path, bound through the C ABI of
--   libsla_hip.so (include/sla_hip.h).  Re-exports the reference's names so callers only change an import.
--   NOT compiled in the authoring image (no GHC there); see haskell/README.md.
module Numeric.LinearAlgebra.Sparse.HIP
  ( linSolve0, LinSolveMethod(..), (#>), (<.>), norm2, arnoldi, (<\>), triLowerSolve, triUpperSolve
  , BICGSTAB, bicgsInit, bicgstabStep, _xBicgstab, _rBicgstab, _pBicgstab
  , CGS, cgsInit, cgsStep, _x, _r, _p, _u
  ) where

import Control.Monad.Catch (MonadThrow, throwM)
import Control.Monad.IO.Class (MonadIO, liftIO)
import Data.Int (Int64)
import Foreign
import Foreign.C.String
import Foreign.C.Types
import System.IO.Unsafe (unsafePerformIO)

import Control.Exception.Common (IterationException (..), MatrixException (..), OperandSizeMismatch (..))
import qualified Data.Sparse.SpMatrix as R
import qualified Data.Sparse.SpVector as R
import Numeric.LinearAlgebra.Sparse (LinSolveMethod (..))

data Ctx; data Csr; data Vec; data Solver

foreign import ccall safe "sla_ctx_create"      c_ctx_create      :: CInt -> Ptr (Ptr Ctx) -> IO CInt
foreign import ccall safe "sla_csr_from_coo"    c_csr_from_coo    :: Ptr Ctx -> Int64 -> Int64 -> Int64 -> Ptr Int64 -> Ptr Int64 -> Ptr Double -> CInt -> Ptr (Ptr Csr) -> IO CInt
foreign import ccall safe "&sla_csr_destroy"    p_csr_destroy     :: FunPtr (Ptr Csr -> IO ())
foreign import ccall safe "sla_vec_create"      c_vec_create      :: Ptr Ctx -> Int64 -> Ptr Double -> Ptr (Ptr Vec) -> IO CInt
foreign import ccall safe "&sla_vec_destroy"    p_vec_destroy     :: FunPtr (Ptr Vec -> IO ())
foreign import ccall safe "sla_vec_to_host"     c_vec_to_host     :: Ptr Vec -> Ptr Double -> IO CInt
foreign import ccall safe "sla_spmv"            c_spmv            :: Ptr Csr -> Ptr Vec -> Ptr Vec -> IO CInt
foreign import ccall safe "sla_dot"             c_dot             :: Ptr Vec -> Ptr Vec -> Ptr Double -> IO CInt
foreign import ccall safe "sla_nrm2"            c_nrm2            :: Ptr Vec -> Ptr Double -> IO CInt
foreign import ccall safe "sla_solver_init"     c_solver_init     :: CInt -> Ptr Csr -> Ptr Vec -> Ptr Vec -> Ptr (Ptr Solver) -> IO CInt
foreign import ccall safe "sla_solver_step"     c_solver_step     :: Ptr Solver -> CInt -> IO CInt
foreign import ccall safe "sla_solver_get"      c_solver_get      :: Ptr Solver -> CInt -> Ptr Vec -> IO CInt
foreign import ccall safe "&sla_solver_destroy" p_solver_destroy  :: FunPtr (Ptr Solver -> IO ())
foreign import ccall safe "sla_linsolve0"       c_linsolve0       :: CInt -> Ptr Csr -> Ptr Vec -> Ptr Vec -> Ptr () -> Ptr Vec -> Ptr () -> IO CInt
foreign import ccall safe "sla_arnoldi"         c_arnoldi         :: Ptr Csr -> Ptr Vec -> CInt -> Ptr Double -> Ptr Double -> Ptr CInt -> IO CInt
foreign import ccall safe "sla_linsolve"        c_linsolve        :: Ptr Csr -> Ptr Vec -> Ptr Vec -> Ptr () -> IO CInt
foreign import ccall safe "sla_tri_solve"       c_tri_solve       :: Ptr Csr -> CInt -> Ptr Vec -> Ptr Vec -> Ptr Int64 -> IO CInt
foreign import ccall unsafe "sla_last_error"    c_last_error      :: IO CString

{-# NOINLINE defaultCtx #-}
defaultCtx :: Ptr Ctx
defaultCtx = unsafePerformIO $ alloca $ \p -> c_ctx_create 0 p >>= check "sla_ctx_create" >> peek p

-- | status code -> the reference's exception / error (Control/Exception/Common.hs:44-76)
check :: String -> CInt -> IO ()
check _ 0 = return ()
check who 1 = c_last_error >>= peekCString >>= \s -> throwM (MatVecSizeMismatchException (who ++ " : " ++ s) (0, 0) 0)
check who 2 = throwM (IterE who "Only BICGSTAB_, CGS_, and CGNE_ are implemented" :: IterationException ())
check _ 3 = error "insertSpMatrix : index out of bounds"
check who 9 = c_last_error >>= peekCString >>= \s -> throwM (NeedsPivoting who s :: MatrixException ())
check who _ = c_last_error >>= peekCString >>= \s -> ioError (userError (who ++ ": " ++ s))

-- | fromListSM semantics are re-applied by the library (sort, last duplicate wins); toListSM's descending
--   order is irrelevant.  A production shim memoises this per SpMatrix (StableName -> ForeignPtr Csr).
lower :: R.SpMatrix Double -> IO (ForeignPtr Csr)
lower aa =
  withArrayLen is $ \nnz pr -> withArray js $ \pc -> withArray xs $ \pv -> alloca $ \out -> do
    c_csr_from_coo defaultCtx (fromIntegral m) (fromIntegral n) (fromIntegral nnz) pr pc pv 0 out >>= check "fromListSM"
    peek out >>= newForeignPtr p_csr_destroy
  where
    (m, n) = R.dim aa
    (is, js, xs) = unzip3 [(fromIntegral i, fromIntegral j, x) | (i, j, x) <- R.toListSM aa]

upload :: R.SpVector Double -> IO (ForeignPtr Vec)
upload v = withArray (R.toDenseListSV v) $ \p -> alloca $ \out -> do
  c_vec_create defaultCtx (fromIntegral (R.dim v)) p out >>= check "sla_vec_create"
  peek out >>= newForeignPtr p_vec_destroy

zeros :: Int -> IO (ForeignPtr Vec)
zeros n = alloca $ \out -> c_vec_create defaultCtx (fromIntegral n) nullPtr out >>= check "sla_vec_create" >> peek out >>= newForeignPtr p_vec_destroy

download :: Int -> ForeignPtr Vec -> IO (R.SpVector Double)
download n fv = withForeignPtr fv $ \v -> allocaArray n $ \p -> do
  c_vec_to_host v p >>= check "sla_vec_to_host"
  R.fromListDenseSV n <$> peekArray n p

-- | linSolve0 (Sparse.hs:1016-1072)
linSolve0 :: (MonadThrow m, MonadIO m) => LinSolveMethod -> R.SpMatrix Double -> R.SpVector Double -> R.SpVector Double -> m (R.SpVector Double)
linSolve0 method aa b x0 = liftIO $ do
  a <- lower aa; vb <- upload b; vx <- upload x0; vo <- zeros (R.ncols aa)
  withForeignPtr a $ \pa -> withForeignPtr vb $ \pb -> withForeignPtr vx $ \px -> withForeignPtr vo $ \po ->
    c_linsolve0 (fromIntegral (fromEnum method)) pa pb px nullPtr po nullPtr >>= check "linSolve0"
  download (R.ncols aa) vo

-- | (#>) (Common.hs:242-250): keys of the result = rows present in the matrix
(#>) :: R.SpMatrix Double -> R.SpVector Double -> R.SpVector Double
aa #> x = unsafePerformIO $ do
  a <- lower aa; vx <- upload x; vy <- zeros (R.nrows aa)
  withForeignPtr a $ \pa -> withForeignPtr vx $ \px -> withForeignPtr vy $ \py -> c_spmv pa px py >>= check "matVec"
  y <- download (R.nrows aa) vy
  return (R.fromListSV (R.nrows aa) [(i, yi) | (i, yi) <- R.toListSV y, i `elem` rowKeys])
  where rowKeys = [i | (i, _, _) <- R.toListSM aa]

(<.>) :: R.SpVector Double -> R.SpVector Double -> Double
v <.> w = unsafePerformIO $ do
  a <- upload v; b <- upload w
  withForeignPtr a $ \pa -> withForeignPtr b $ \pb -> alloca $ \out -> c_dot pa pb out >>= check "<.>" >> peek out

norm2 :: R.SpVector Double -> Double
norm2 v = unsafePerformIO $ upload v >>= \a -> withForeignPtr a $ \pa -> alloca $ \out -> c_nrm2 pa out >>= check "norm2" >> peek out

-- | solver state records: the device keeps x, r, p (, u); field accessors download on demand
newtype BICGSTAB = BICGSTAB (ForeignPtr Solver, Int)
newtype CGS = CGS (ForeignPtr Solver, Int)

initWith :: CInt -> R.SpMatrix Double -> R.SpVector Double -> R.SpVector Double -> IO (ForeignPtr Solver)
initWith meth aa b x0 = do
  a <- lower aa; vb <- upload b; vx <- upload x0
  withForeignPtr a $ \pa -> withForeignPtr vb $ \pb -> withForeignPtr vx $ \px -> alloca $ \out -> do
    c_solver_init meth pa pb px out >>= check "solver init"
    peek out >>= newForeignPtr p_solver_destroy

field :: CInt -> (ForeignPtr Solver, Int) -> R.SpVector Double
field k (fs, n) = unsafePerformIO $ do
  v <- zeros n
  withForeignPtr fs $ \s -> withForeignPtr v $ \pv -> c_solver_get s k pv >>= check "solver get"
  download n v

bicgsInit :: R.SpMatrix Double -> R.SpVector Double -> R.SpVector Double -> BICGSTAB
bicgsInit aa b x0 = BICGSTAB (unsafePerformIO (initWith 4 aa b x0), R.ncols aa)

-- | k applications of bicgstabStep (Sparse.hs:972-981); the shadow residual lives in the device state
bicgstabStep :: Int -> BICGSTAB -> BICGSTAB
bicgstabStep k s@(BICGSTAB (fs, _)) = unsafePerformIO $ withForeignPtr fs (\p -> c_solver_step p (fromIntegral k) >>= check "bicgstabStep") >> return s

_xBicgstab, _rBicgstab, _pBicgstab :: BICGSTAB -> R.SpVector Double
_xBicgstab (BICGSTAB s) = field 0 s; _rBicgstab (BICGSTAB s) = field 1 s; _pBicgstab (BICGSTAB s) = field 2 s

cgsInit :: R.SpMatrix Double -> R.SpVector Double -> R.SpVector Double -> CGS
cgsInit aa b x0 = CGS (unsafePerformIO (initWith 3 aa b x0), R.ncols aa)

cgsStep :: Int -> CGS -> CGS
cgsStep k s@(CGS (fs, _)) = unsafePerformIO $ withForeignPtr fs (\p -> c_solver_step p (fromIntegral k) >>= check "cgsStep") >> return s

_x, _r, _p, _u :: CGS -> R.SpVector Double
_x (CGS s) = field 0 s; _r (CGS s) = field 1 s; _p (CGS s) = field 2 s; _u (CGS s) = field 3 s

-- | arnoldi (Sparse.hs:630-667): Q n x (k+1), H (k+1) x k
arnoldi :: (MonadThrow m, MonadIO m) => R.SpMatrix Double -> R.SpVector Double -> Int -> m (R.SpMatrix Double, R.SpMatrix Double)
arnoldi aa b kn = liftIO $ do
  a <- lower aa; vb <- upload b
  let n = R.ncols aa
  allocaArray (n * (kn + 1)) $ \pq -> allocaArray ((kn + 1) * kn) $ \ph -> alloca $ \pk ->
    withForeignPtr a $ \pa -> withForeignPtr vb $ \pb -> do
      c_arnoldi pa pb (fromIntegral kn) pq ph pk >>= check "arnoldi"
      k <- fromIntegral <$> peek pk
      q <- peekArray (n * (k + 1)) pq
      h <- peekArray ((kn + 1) * kn) ph
      return ( R.fromListDenseSM n q
             , R.fromListSM (k + 1, k) [(i, j, h !! (j * (kn + 1) + i)) | j <- [0 .. k - 1], i <- [0 .. j + 1]] )

-- | (<\>) (Class.hs:244-249) as the dead instance defined it (Sparse.hs:1080-1084)
(<\>) :: (MonadThrow m, MonadIO m) => R.SpMatrix Double -> R.SpVector Double -> m (R.SpVector Double)
aa <\> b = liftIO $ do
  a <- lower aa; vb <- upload b; vo <- zeros (R.ncols aa)
  withForeignPtr a $ \pa -> withForeignPtr vb $ \pb -> withForeignPtr vo $ \po -> c_linsolve pa pb po nullPtr >>= check "<\\>"
  download (R.ncols aa) vo

-- | triLowerSolve / triUpperSolve (Sparse.hs:750-811): status 9 is the reference's NeedsPivoting (mapped in `check`);
--   the device result is already sparsifySV-ed
triLowerSolve, triUpperSolve :: (MonadThrow m, MonadIO m) => R.SpMatrix Double -> R.SpVector Double -> m (R.SpVector Double)
triLowerSolve = triSolve 0 "triLowerSolve"
triUpperSolve = triSolve 1 "triUpperSolve"

triSolve :: (MonadThrow m, MonadIO m) => CInt -> String -> R.SpMatrix Double -> R.SpVector Double -> m (R.SpVector Double)
triSolve upper who tt b = liftIO $ do
  t <- lower tt; vb <- upload b; vo <- zeros (R.nrows tt)
  withForeignPtr t $ \pt -> withForeignPtr vb $ \pb -> withForeignPtr vo $ \po -> c_tri_solve pt upper pb po nullPtr >>= check who
  R.sparsifySV <$> download (R.nrows tt) vo
